// Sanitizer harness of the host layer's concurrent database use (tests/test_host_layer_cpu.py builds it with
// -fsanitize=thread and with -fsanitize=address,undefined and runs it): what RunGrouped does - one writer thread
// per group inside a DatabaseTransaction (deletes + inserts of both tables) while the calling thread already
// filters and reads the next group through the same connection - plus a transaction that is abandoned by an
// exception (must roll back, must not terminate).
#include <cstdio>
#include <future>
#include <stdexcept>
#include <vector>

#include "../../pycolmap_amd/csrc/host/database.h"

using namespace amchost;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    Database db(argv[1]);  // creates the file and the schema
    db.SetBulkWriteMode(true);
    const int groups = 12, per_group = 40;
    std::future<void> writer;
    size_t reads = 0;
    for (int g = 0; g < groups; ++g) {
        // "Compute": the main thread reads rows of earlier groups while the previous group is being written
        for (int k = 0; k < per_group; ++k) {
            const image_t a = 1 + (g * per_group + k) % 97, b = a + 1 + k % 5;
            reads += db.ExistsMatches(a, b) ? db.ReadMatches(a, b).size() : 0;
            reads += db.ReadMatchedPairIds().size();
        }
        if (writer.valid()) writer.get();
        writer = std::async(std::launch::async, [&db, g] {
            DatabaseTransaction tx(&db);
            for (int k = 0; k < per_group; ++k) {
                const image_t a = 1 + (g * per_group + k) % 97, b = a + 1 + k % 5;
                std::vector<uint32_t> m(2 * (10 + k), static_cast<uint32_t>(g));
                if (db.ExistsMatches(a, b)) db.DeleteMatches(a, b);
                if (db.ExistsInlierMatches(a, b)) db.DeleteInlierMatches(a, b);
                db.WriteMatches(a, b, m);
                TwoViewGeometryRow t;
                t.config = 2;
                t.inlier_matches = m;
                t.H = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
                db.WriteTwoViewGeometry(a, b, t);
            }
            tx.Commit();
        });
    }
    writer.get();
    const size_t rows = db.NumMatchedImagePairs();
    // a transaction left by an exception rolls back and does not take the process down
    try {
        DatabaseTransaction tx(&db);
        db.WriteMatches(500, 501, std::vector<uint32_t>{1, 2});
        throw std::runtime_error("boom");
    } catch (const std::runtime_error&) {
    }
    if (db.ExistsMatches(500, 501)) {
        std::fprintf(stderr, "rollback failed\n");
        return 1;
    }
    std::printf("ok rows=%zu reads=%zu\n", rows, reads);
    return rows > 0 ? 0 : 1;
}
