// database.h — minimal C++ access to a COLMAP SQLite database (schema: SURVEY.md A.5, which
// mirrors COLMAP's scripts/python/database.py).  Only what the match + verify path touches:
// cameras, images, keypoints, descriptors, matches, two_view_geometries.
// Python surface mirrored: /root/reference/pycolmap/scene/database.h:9-46.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

struct sqlite3;
struct sqlite3_stmt;

namespace amchost {

using image_t = uint32_t;
using camera_t = uint32_t;
using image_pair_t = uint64_t;

constexpr uint64_t kMaxNumImages = 2147483647ull;  // COLMAP Database::kMaxNumImages

struct CameraRow {
    camera_t camera_id = 0;
    int model_id = 0;
    uint64_t width = 0, height = 0;
    std::vector<double> params;
    bool has_prior_focal_length = false;
};
struct ImageRow {
    image_t image_id = 0;
    std::string name;
    camera_t camera_id = 0;
    // Image::TvecPrior: prior_tx / ty / tz (latitude, longitude, altitude when the priors are GPS coordinates);
    // a NULL column reads as NaN, like COLMAP's Database::ReadImage
    std::array<double, 3> prior_t{{std::nan(""), std::nan(""), std::nan("")}};
    std::array<double, 4> prior_q{{std::nan(""), std::nan(""), std::nan(""), std::nan("")}};  // prior_qw, qx, qy, qz
};
struct TwoViewGeometryRow {
    int config = 0;  // UNDEFINED
    std::array<double, 9> F{}, E{}, H{};  // row-major
    std::vector<uint32_t> inlier_matches;  // rows x 2
    std::array<double, 4> qvec{{1, 0, 0, 0}};
    std::array<double, 3> tvec{{0, 0, 0}};
    void Invert();  // TwoViewGeometry::Invert: swap roles of the two images
};

class Database {
  public:
    explicit Database(const std::string& path);
    ~Database();
    Database(const Database&) = delete;
    Database& operator=(const Database&) = delete;

    // pair id helpers (COLMAP Database::ImagePairToPairId / SwapImagePair / PairIdToImagePair)
    static bool SwapImagePair(image_t id1, image_t id2) { return id1 > id2; }
    static image_pair_t ImagePairToPairId(image_t id1, image_t id2);
    static void PairIdToImagePair(image_pair_t pair_id, image_t* id1, image_t* id2);

    size_t NumCameras() const { return Count("cameras"); }
    size_t NumImages() const { return Count("images"); }
    size_t NumKeypoints() const { return SumRows("keypoints"); }
    size_t NumDescriptors() const { return SumRows("descriptors"); }
    size_t NumMatches() const { return SumRows("matches"); }
    size_t NumInlierMatches() const { return SumRows("two_view_geometries"); }
    size_t NumMatchedImagePairs() const { return Count("matches"); }
    size_t NumVerifiedImagePairs() const { return Count("two_view_geometries"); }

    std::vector<CameraRow> ReadAllCameras() const;
    std::vector<ImageRow> ReadAllImages() const;  // ordered by image_id
    // Database::ExistsCamera / ExistsImage / ExistsImageWithName, ReadCamera / ReadImage / ReadImageWithName (a missing
    // row throws std::invalid_argument), WriteCamera / WriteImage (the id is SQLite's unless use_*_id; returns it),
    // NumKeypointsForImage / NumDescriptorsForImage (0 without a row)
    bool ExistsCamera(camera_t camera_id) const;
    bool ExistsImage(image_t image_id) const;
    bool ExistsImageWithName(const std::string& name) const;
    CameraRow ReadCamera(camera_t camera_id) const;
    ImageRow ReadImage(image_t image_id) const;
    ImageRow ReadImageWithName(const std::string& name) const;
    camera_t WriteCamera(const CameraRow& camera, bool use_camera_id = false);
    image_t WriteImage(const ImageRow& image, bool use_image_id = false);
    size_t NumKeypointsForImage(image_t image_id) const { return RowsOf("keypoints", image_id); }
    size_t NumDescriptorsForImage(image_t image_id) const { return RowsOf("descriptors", image_id); }
    // Database::Open / Close: Close() finishes the prepared statements and the connection (idempotent); Open() closes
    // what is open and opens `path`, creating COLMAP's tables when they are missing
    void Open(const std::string& path);
    void Close();
    // keypoints blob: rows x cols float32; returns x,y only (rows x 2)
    std::vector<float> ReadKeypointsXY(image_t image_id, uint32_t* rows) const;
    std::vector<uint8_t> ReadDescriptors(image_t image_id, uint32_t* rows) const;
    // the whole keypoints blob (rows x cols float32, cols = 2, 4 or 6)
    std::vector<float> ReadKeypoints(image_t image_id, uint32_t* rows, uint32_t* cols) const;
    // Database::WriteKeypoints / WriteDescriptors: one row per image, an existing row is an error
    void WriteKeypoints(image_t image_id, const float* data, uint32_t rows, uint32_t cols);
    void WriteDescriptors(image_t image_id, const uint8_t* data, uint32_t rows);
    bool ExistsKeypoints(image_t image_id) const;
    bool ExistsDescriptors(image_t image_id) const;

    // all pair ids that have a row in `matches` / `two_view_geometries` (one scan instead of two point
    // queries per candidate pair)
    std::vector<image_pair_t> ReadMatchedPairIds() const;
    std::vector<image_pair_t> ReadVerifiedPairIds() const;
    bool ExistsMatches(image_t id1, image_t id2) const;
    bool ExistsInlierMatches(image_t id1, image_t id2) const;
    // matches as stored for the ordered pair (id1, id2): columns swapped back if id1 > id2
    std::vector<uint32_t> ReadMatches(image_t id1, image_t id2) const;
    TwoViewGeometryRow ReadTwoViewGeometry(image_t id1, image_t id2) const;
    void WriteMatches(image_t id1, image_t id2, const std::vector<uint32_t>& matches);
    void WriteTwoViewGeometry(image_t id1, image_t id2, const TwoViewGeometryRow& tvg);
    void DeleteMatches(image_t id1, image_t id2);
    void DeleteInlierMatches(image_t id1, image_t id2);

    void BeginTransaction();
    void EndTransaction();
    void RollbackTransaction() noexcept;
    // Bulk-write mode of the matching controllers.  COLMAP opens its databases with journal_mode=WAL
    // (kept: it is what a reader of the file finds afterwards), but under WAL every page of the
    // gigabytes of match blobs a run appends is written twice (log, then checkpoint).  While a
    // controller owns the connection the journal is a rollback journal (TRUNCATE: appended pages are
    // written once, only overwritten pages are journalled); WAL is restored when the mode is switched
    // off or the database is closed.  If another connection holds the file SQLite refuses the switch
    // and the run simply stays in WAL.  Returns the journal mode in effect.
    std::string SetBulkWriteMode(bool on);

  private:
    void CreateTables() const;
    size_t Count(const char* table) const;
    size_t SumRows(const char* table) const;
    size_t RowsOf(const char* table, image_t image_id) const;
    bool ExistsPair(const char* table, image_pair_t pair_id) const;
    void Exec(const char* sql) const;
    sqlite3_stmt* Prepared(const std::string& sql) const;  // prepared once per connection, then reused
    sqlite3* db_ = nullptr;
    bool bulk_mode_ = false;
    // every public call holds this: the matching controllers write one group from a worker thread
    // while the main thread already filters the next group against the same connection
    mutable std::recursive_mutex mu_;
    mutable std::unordered_map<std::string, sqlite3_stmt*> stmts_;
};

// BEGIN on construction; Commit() ends the transaction, and a scope left without it (an exception on the
// way) rolls back: a group whose write failed half way leaves no partial rows, and the destructor never throws.
class DatabaseTransaction {
  public:
    explicit DatabaseTransaction(Database* db) : db_(db) { db_->BeginTransaction(); }
    void Commit() {
        done_ = true;
        db_->EndTransaction();
    }
    ~DatabaseTransaction() {
        if (!done_) db_->RollbackTransaction();
    }
    DatabaseTransaction(const DatabaseTransaction&) = delete;
    DatabaseTransaction& operator=(const DatabaseTransaction&) = delete;
  private:
    Database* db_;
    bool done_ = false;
};

}  // namespace amchost
