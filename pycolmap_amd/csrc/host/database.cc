#include "database.h"

#include <sqlite3.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace amchost {

namespace {
[[noreturn]] void Fail(sqlite3* db, const std::string& what) {
    throw std::runtime_error("SQLite error (" + what + "): " + (db ? sqlite3_errmsg(db) : "no db"));
}
// A prepared statement borrowed from the connection's cache for one use: the per-pair statements
// (existence checks, match / two-view-geometry inserts) run hundreds of thousands of times per job,
// so they are compiled once and reset + re-bound afterwards.
struct Stmt {
    sqlite3_stmt* s = nullptr;
    sqlite3* db;
    Stmt(sqlite3* d, sqlite3_stmt* cached) : s(cached), db(d) {}
    ~Stmt() {
        sqlite3_reset(s);
        sqlite3_clear_bindings(s);
    }
    Stmt(const Stmt&) = delete;
    Stmt& operator=(const Stmt&) = delete;
    bool Step() {
        const int rc = sqlite3_step(s);
        if (rc == SQLITE_ROW) return true;
        if (rc == SQLITE_DONE) return false;
        Fail(db, "step");
    }
};
}  // namespace

// TwoViewGeometry::Invert (colmap/scene/two_view_geometry.cc): F <- F^T, E <- E^T, H <- H^-1,
// swap the match columns, cam2_from_cam1 <- Inverse(cam2_from_cam1) (colmap/geometry/rigid3.h: the
// inverse quaternion, then the rotated negated translation, both as Eigen evaluates them).
void TwoViewGeometryRow::Invert() {
    auto transpose = [](std::array<double, 9>& m) {
        std::swap(m[1], m[3]);
        std::swap(m[2], m[6]);
        std::swap(m[5], m[7]);
    };
    transpose(F);
    transpose(E);
    const std::array<double, 9> h = H;
    const double c00 = h[4] * h[8] - h[5] * h[7], c01 = h[5] * h[6] - h[3] * h[8], c02 = h[3] * h[7] - h[4] * h[6];
    const double det = h[0] * c00 + h[1] * c01 + h[2] * c02;
    {   // Eigen's 3 x 3 inverse: cofactors / determinant; a singular H gives inf / NaN entries there too
        const double inv = 1.0 / det;
        H[0] = c00 * inv; H[1] = (h[2] * h[7] - h[1] * h[8]) * inv; H[2] = (h[1] * h[5] - h[2] * h[4]) * inv;
        H[3] = c01 * inv; H[4] = (h[0] * h[8] - h[2] * h[6]) * inv; H[5] = (h[2] * h[3] - h[0] * h[5]) * inv;
        H[6] = c02 * inv; H[7] = (h[1] * h[6] - h[0] * h[7]) * inv; H[8] = (h[0] * h[4] - h[1] * h[3]) * inv;
    }
    for (size_t i = 0; i + 1 < inlier_matches.size(); i += 2) std::swap(inlier_matches[i], inlier_matches[i + 1]);
    {
        // Eigen::Quaterniond::inverse(): conjugate / squaredNorm (qvec is stored w, x, y, z)
        const double x = qvec[1], y = qvec[2], z = qvec[3], w = qvec[0];
        const double n2 = x * x + y * y + z * z + w * w;
        double iw = 0.0, ix = 0.0, iy = 0.0, iz = 0.0;
        if (n2 > 0.0) { ix = -x / n2; iy = -y / n2; iz = -z / n2; iw = w / n2; }
        // Eigen's quaternion * vector: v + w * (2 q x v) + q x (2 q x v), v = -t
        const double v0 = -tvec[0], v1 = -tvec[1], v2 = -tvec[2];
        double u0 = iy * v2 - iz * v1, u1 = iz * v0 - ix * v2, u2 = ix * v1 - iy * v0;
        u0 += u0; u1 += u1; u2 += u2;
        const double r0 = v0 + iw * u0 + (iy * u2 - iz * u1);
        const double r1 = v1 + iw * u1 + (iz * u0 - ix * u2);
        const double r2 = v2 + iw * u2 + (ix * u1 - iy * u0);
        qvec = {{iw, ix, iy, iz}};
        tvec = {{r0, r1, r2}};
    }
}

Database::Database(const std::string& path) { Open(path); }
void Database::Open(const std::string& path) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Close();
    // Database::Open: the file and COLMAP's tables are created when missing
    if (sqlite3_open_v2(path.c_str(), &db_, SQLITE_OPEN_READWRITE | SQLITE_OPEN_CREATE, nullptr) != SQLITE_OK) {
        const std::string msg = db_ ? sqlite3_errmsg(db_) : "out of memory";
        sqlite3_close(db_);
        db_ = nullptr;
        throw std::runtime_error("cannot open database " + path + ": " + msg);
    }
    Exec("PRAGMA synchronous=OFF");       // as COLMAP's Database::Open
    Exec("PRAGMA journal_mode=WAL");
    Exec("PRAGMA foreign_keys=ON");
    CreateTables();
}
void Database::Close() {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    if (db_ && bulk_mode_) {
        try {
            SetBulkWriteMode(false);
        } catch (...) {
        }
    }
    for (auto& kv : stmts_) sqlite3_finalize(kv.second);
    stmts_.clear();
    if (db_) sqlite3_close(db_);
    db_ = nullptr;
    bulk_mode_ = false;
}
// Database::CreateTables (colmap/scene/database.cc): CREATE TABLE IF NOT EXISTS for every table of the schema
void Database::CreateTables() const {
    Exec("CREATE TABLE IF NOT EXISTS cameras (camera_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, "
         "model INTEGER NOT NULL, width INTEGER NOT NULL, height INTEGER NOT NULL, params BLOB, "
         "prior_focal_length INTEGER NOT NULL);");
    Exec("CREATE TABLE IF NOT EXISTS images (image_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, "
         "name TEXT NOT NULL UNIQUE, camera_id INTEGER NOT NULL, prior_qw REAL, prior_qx REAL, prior_qy REAL, "
         "prior_qz REAL, prior_tx REAL, prior_ty REAL, prior_tz REAL, "
         "CONSTRAINT image_id_check CHECK(image_id >= 0 and image_id < 2147483647), "
         "FOREIGN KEY(camera_id) REFERENCES cameras(camera_id));");
    Exec("CREATE UNIQUE INDEX IF NOT EXISTS index_name ON images(name);");
    Exec("CREATE TABLE IF NOT EXISTS keypoints (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, "
         "cols INTEGER NOT NULL, data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);");
    Exec("CREATE TABLE IF NOT EXISTS descriptors (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, "
         "cols INTEGER NOT NULL, data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);");
    Exec("CREATE TABLE IF NOT EXISTS matches (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, "
         "cols INTEGER NOT NULL, data BLOB);");
    Exec("CREATE TABLE IF NOT EXISTS two_view_geometries (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, "
         "cols INTEGER NOT NULL, data BLOB, config INTEGER NOT NULL, F BLOB, E BLOB, H BLOB, qvec BLOB, tvec BLOB);");
}
Database::~Database() { Close(); }
std::string Database::SetBulkWriteMode(bool on) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    if (!db_) throw std::runtime_error("the database is closed");
    // cached statements hold the schema; a journal-mode switch needs no statement in progress (all are reset)
    sqlite3_stmt* st = nullptr;
    const char* sql = on ? "PRAGMA journal_mode=TRUNCATE" : "PRAGMA journal_mode=WAL";
    if (sqlite3_prepare_v2(db_, sql, -1, &st, nullptr) != SQLITE_OK) Fail(db_, sql);
    std::string mode;
    if (sqlite3_step(st) == SQLITE_ROW) {
        const unsigned char* t = sqlite3_column_text(st, 0);
        if (t) mode = reinterpret_cast<const char*>(t);
    }
    sqlite3_finalize(st);
    bulk_mode_ = on && mode == "truncate";
    // connection-level settings for the append phase (they do not change a byte of the file): by default none;
    // AMC_DB_BULK_PRAGMAS="mmap_size=17179869184;locking_mode=EXCLUSIVE" (semicolon-separated, without the PRAGMA
    // keyword) tries others - tools/pipeline_bench.py is how they are measured.  Switched back when the mode ends.
    if (const char* e = std::getenv("AMC_DB_BULK_PRAGMAS")) {
        const std::string list(e);
        size_t p0 = 0;
        while (p0 < list.size()) {
            size_t q = list.find(';', p0);
            if (q == std::string::npos) q = list.size();
            std::string one = list.substr(p0, q - p0);
            p0 = q + 1;
            if (one.empty()) continue;
            if (!on) {  // back to SQLite's defaults for the settings that stick to the connection
                const std::string key = one.substr(0, one.find('='));
                if (key == "locking_mode") one = "locking_mode=NORMAL";
                else if (key == "mmap_size") one = "mmap_size=0";
                else if (key == "cache_size") one = "cache_size=-2000";
                else continue;
            }
            try {
                Exec(("PRAGMA " + one).c_str());
            } catch (...) {
            }
        }
    }
    return mode;
}
sqlite3_stmt* Database::Prepared(const std::string& sql) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    if (!db_) throw std::runtime_error("the database is closed");
    auto it = stmts_.find(sql);
    if (it != stmts_.end()) return it->second;
    sqlite3_stmt* st = nullptr;
    if (sqlite3_prepare_v2(db_, sql.c_str(), -1, &st, nullptr) != SQLITE_OK) Fail(db_, sql);
    stmts_.emplace(sql, st);
    return st;
}
void Database::Exec(const char* sql) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    if (!db_) throw std::runtime_error("the database is closed");
    char* err = nullptr;
    if (sqlite3_exec(db_, sql, nullptr, nullptr, &err) != SQLITE_OK) {
        const std::string msg = err ? err : "?";
        sqlite3_free(err);
        throw std::runtime_error(std::string("SQLite exec failed: ") + sql + ": " + msg);
    }
}
void Database::BeginTransaction() {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Exec("BEGIN TRANSACTION");
}
void Database::EndTransaction() {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Exec("END TRANSACTION");
}
void Database::RollbackTransaction() noexcept {
    try {
        std::lock_guard<std::recursive_mutex> lock(mu_);
        for (auto& kv : stmts_) sqlite3_reset(kv.second);  // a statement left mid-step would block the rollback
        Exec("ROLLBACK TRANSACTION");
    } catch (...) {
    }
}

image_pair_t Database::ImagePairToPairId(image_t id1, image_t id2) {
    if (id1 >= kMaxNumImages || id2 >= kMaxNumImages) throw std::invalid_argument("image_id out of range");
    if (SwapImagePair(id1, id2)) return kMaxNumImages * id2 + id1;
    return kMaxNumImages * id1 + id2;
}
void Database::PairIdToImagePair(image_pair_t pair_id, image_t* id1, image_t* id2) {
    *id2 = static_cast<image_t>(pair_id % kMaxNumImages);
    *id1 = static_cast<image_t>((pair_id - *id2) / kMaxNumImages);
}

size_t Database::Count(const char* table) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared((std::string("SELECT COUNT(*) FROM ") + table)));
    st.Step();
    return static_cast<size_t>(sqlite3_column_int64(st.s, 0));
}
size_t Database::SumRows(const char* table) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared((std::string("SELECT SUM(rows) FROM ") + table)));
    st.Step();
    return static_cast<size_t>(sqlite3_column_int64(st.s, 0));
}

namespace {
constexpr const char* kCameraCols = "camera_id, model, width, height, params, prior_focal_length";
constexpr const char* kImageCols = "image_id, name, camera_id, prior_qw, prior_qx, prior_qy, prior_qz, prior_tx, prior_ty, prior_tz";
CameraRow CameraFromRow(sqlite3_stmt* st) {
    CameraRow c;
    c.camera_id = static_cast<camera_t>(sqlite3_column_int64(st, 0));
    c.model_id = sqlite3_column_int(st, 1);
    c.width = static_cast<uint64_t>(sqlite3_column_int64(st, 2));
    c.height = static_cast<uint64_t>(sqlite3_column_int64(st, 3));
    const int nbytes = sqlite3_column_bytes(st, 4);
    c.params.resize(nbytes / sizeof(double));
    if (nbytes) std::memcpy(c.params.data(), sqlite3_column_blob(st, 4), c.params.size() * sizeof(double));
    c.has_prior_focal_length = sqlite3_column_int(st, 5) != 0;
    return c;
}
ImageRow ImageFromRow(sqlite3_stmt* st) {  // Database::ReadImageRow: NULL prior columns read as NaN
    ImageRow r;
    r.image_id = static_cast<image_t>(sqlite3_column_int64(st, 0));
    r.name = reinterpret_cast<const char*>(sqlite3_column_text(st, 1));
    r.camera_id = static_cast<camera_t>(sqlite3_column_int64(st, 2));
    for (int k = 0; k < 4; ++k)
        if (sqlite3_column_type(st, 3 + k) != SQLITE_NULL) r.prior_q[k] = sqlite3_column_double(st, 3 + k);
    for (int k = 0; k < 3; ++k)
        if (sqlite3_column_type(st, 7 + k) != SQLITE_NULL) r.prior_t[k] = sqlite3_column_double(st, 7 + k);
    return r;
}
}  // namespace

std::vector<CameraRow> Database::ReadAllCameras() const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    std::vector<CameraRow> out;
    Stmt st(db_, Prepared(std::string("SELECT ") + kCameraCols + " FROM cameras ORDER BY camera_id"));
    while (st.Step()) out.push_back(CameraFromRow(st.s));
    return out;
}
std::vector<ImageRow> Database::ReadAllImages() const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    std::vector<ImageRow> out;
    Stmt st(db_, Prepared(std::string("SELECT ") + kImageCols + " FROM images ORDER BY image_id"));
    while (st.Step()) out.push_back(ImageFromRow(st.s));
    return out;
}
bool Database::ExistsCamera(camera_t camera_id) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("SELECT 1 FROM cameras WHERE camera_id = ?"));
    sqlite3_bind_int64(st.s, 1, camera_id);
    return st.Step();
}
bool Database::ExistsImage(image_t image_id) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("SELECT 1 FROM images WHERE image_id = ?"));
    sqlite3_bind_int64(st.s, 1, image_id);
    return st.Step();
}
bool Database::ExistsImageWithName(const std::string& name) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("SELECT 1 FROM images WHERE name = ?"));
    sqlite3_bind_text(st.s, 1, name.c_str(), static_cast<int>(name.size()), SQLITE_STATIC);
    return st.Step();
}
CameraRow Database::ReadCamera(camera_t camera_id) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared(std::string("SELECT ") + kCameraCols + " FROM cameras WHERE camera_id = ?"));
    sqlite3_bind_int64(st.s, 1, camera_id);
    if (!st.Step()) throw std::invalid_argument("camera " + std::to_string(camera_id) + " does not exist");
    return CameraFromRow(st.s);
}
ImageRow Database::ReadImage(image_t image_id) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared(std::string("SELECT ") + kImageCols + " FROM images WHERE image_id = ?"));
    sqlite3_bind_int64(st.s, 1, image_id);
    if (!st.Step()) throw std::invalid_argument("image " + std::to_string(image_id) + " does not exist");
    return ImageFromRow(st.s);
}
ImageRow Database::ReadImageWithName(const std::string& name) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared(std::string("SELECT ") + kImageCols + " FROM images WHERE name = ?"));
    sqlite3_bind_text(st.s, 1, name.c_str(), static_cast<int>(name.size()), SQLITE_STATIC);
    if (!st.Step()) throw std::invalid_argument("image \"" + name + "\" does not exist");
    return ImageFromRow(st.s);
}
// Database::WriteCamera: camera_id NULL (SQLite assigns it) unless use_camera_id, which must not exist yet
camera_t Database::WriteCamera(const CameraRow& c, bool use_camera_id) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    if (use_camera_id && ExistsCamera(c.camera_id))
        throw std::invalid_argument("camera " + std::to_string(c.camera_id) + " exists already");
    Stmt st(db_, Prepared("INSERT INTO cameras(camera_id, model, width, height, params, prior_focal_length) VALUES(?, ?, ?, ?, ?, ?)"));
    if (use_camera_id) sqlite3_bind_int64(st.s, 1, c.camera_id); else sqlite3_bind_null(st.s, 1);
    sqlite3_bind_int64(st.s, 2, c.model_id);
    sqlite3_bind_int64(st.s, 3, static_cast<sqlite3_int64>(c.width));
    sqlite3_bind_int64(st.s, 4, static_cast<sqlite3_int64>(c.height));
    sqlite3_bind_blob(st.s, 5, c.params.data(), static_cast<int>(c.params.size() * sizeof(double)), SQLITE_STATIC);
    sqlite3_bind_int64(st.s, 6, c.has_prior_focal_length ? 1 : 0);
    st.Step();
    return static_cast<camera_t>(sqlite3_last_insert_rowid(db_));
}
// Database::WriteImage: likewise; NaN priors become NULL columns (sqlite3_bind_double stores NaN as NULL)
image_t Database::WriteImage(const ImageRow& im, bool use_image_id) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    if (use_image_id && ExistsImage(im.image_id))
        throw std::invalid_argument("image " + std::to_string(im.image_id) + " exists already");
    Stmt st(db_, Prepared("INSERT INTO images(image_id, name, camera_id, prior_qw, prior_qx, prior_qy, prior_qz, prior_tx, prior_ty, "
                          "prior_tz) VALUES(?, ?, ?, ?, ?, ?, ?, ?, ?, ?)"));
    if (use_image_id) sqlite3_bind_int64(st.s, 1, im.image_id); else sqlite3_bind_null(st.s, 1);
    sqlite3_bind_text(st.s, 2, im.name.c_str(), static_cast<int>(im.name.size()), SQLITE_STATIC);
    sqlite3_bind_int64(st.s, 3, im.camera_id);
    for (int k = 0; k < 4; ++k) sqlite3_bind_double(st.s, 4 + k, im.prior_q[k]);
    for (int k = 0; k < 3; ++k) sqlite3_bind_double(st.s, 8 + k, im.prior_t[k]);
    st.Step();
    return static_cast<image_t>(sqlite3_last_insert_rowid(db_));
}
size_t Database::RowsOf(const char* table, image_t image_id) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared(std::string("SELECT rows FROM ") + table + " WHERE image_id = ?"));
    sqlite3_bind_int64(st.s, 1, image_id);
    return st.Step() ? static_cast<size_t>(sqlite3_column_int64(st.s, 0)) : 0;
}
std::vector<float> Database::ReadKeypointsXY(image_t image_id, uint32_t* rows) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    *rows = 0;
    Stmt st(db_, Prepared("SELECT rows, cols, data FROM keypoints WHERE image_id = ?"));
    sqlite3_bind_int64(st.s, 1, image_id);
    std::vector<float> out;
    if (!st.Step()) return out;
    const uint32_t r = static_cast<uint32_t>(sqlite3_column_int64(st.s, 0));
    const uint32_t c = static_cast<uint32_t>(sqlite3_column_int64(st.s, 1));
    const int nbytes = sqlite3_column_bytes(st.s, 2);
    if (r == 0) return out;
    if (c < 2 || static_cast<size_t>(nbytes) != static_cast<size_t>(r) * c * sizeof(float))
        throw std::runtime_error("keypoints blob of image " + std::to_string(image_id) + " has inconsistent shape");
    const float* src = static_cast<const float*>(sqlite3_column_blob(st.s, 2));
    out.resize(static_cast<size_t>(r) * 2);
    for (uint32_t i = 0; i < r; ++i) {
        out[2 * i] = src[static_cast<size_t>(i) * c];
        out[2 * i + 1] = src[static_cast<size_t>(i) * c + 1];
    }
    *rows = r;
    return out;
}
std::vector<uint8_t> Database::ReadDescriptors(image_t image_id, uint32_t* rows) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    *rows = 0;
    Stmt st(db_, Prepared("SELECT rows, cols, data FROM descriptors WHERE image_id = ?"));
    sqlite3_bind_int64(st.s, 1, image_id);
    std::vector<uint8_t> out;
    if (!st.Step()) return out;
    const uint32_t r = static_cast<uint32_t>(sqlite3_column_int64(st.s, 0));
    const uint32_t c = static_cast<uint32_t>(sqlite3_column_int64(st.s, 1));
    const int nbytes = sqlite3_column_bytes(st.s, 2);
    if (r == 0) return out;
    if (c != 128 || static_cast<size_t>(nbytes) != static_cast<size_t>(r) * 128)
        throw std::runtime_error("descriptors blob of image " + std::to_string(image_id) + " is not rows x 128 uint8");
    out.resize(static_cast<size_t>(nbytes));
    std::memcpy(out.data(), sqlite3_column_blob(st.s, 2), out.size());
    *rows = r;
    return out;
}

std::vector<float> Database::ReadKeypoints(image_t image_id, uint32_t* rows, uint32_t* cols) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    *rows = 0;
    *cols = 0;
    Stmt st(db_, Prepared("SELECT rows, cols, data FROM keypoints WHERE image_id = ?"));
    sqlite3_bind_int64(st.s, 1, image_id);
    std::vector<float> out;
    if (!st.Step()) return out;
    const uint32_t r = static_cast<uint32_t>(sqlite3_column_int64(st.s, 0));
    const uint32_t c = static_cast<uint32_t>(sqlite3_column_int64(st.s, 1));
    const int nbytes = sqlite3_column_bytes(st.s, 2);
    if (static_cast<size_t>(nbytes) != static_cast<size_t>(r) * c * sizeof(float))
        throw std::runtime_error("keypoints blob of image " + std::to_string(image_id) + " has inconsistent shape");
    out.resize(static_cast<size_t>(r) * c);
    if (nbytes) std::memcpy(out.data(), sqlite3_column_blob(st.s, 2), static_cast<size_t>(nbytes));
    *rows = r;
    *cols = c;
    return out;
}
bool Database::ExistsKeypoints(image_t image_id) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("SELECT 1 FROM keypoints WHERE image_id = ?"));
    sqlite3_bind_int64(st.s, 1, image_id);
    return st.Step();
}
bool Database::ExistsDescriptors(image_t image_id) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("SELECT 1 FROM descriptors WHERE image_id = ?"));
    sqlite3_bind_int64(st.s, 1, image_id);
    return st.Step();
}
void Database::WriteKeypoints(image_t image_id, const float* data, uint32_t rows, uint32_t cols) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("INSERT INTO keypoints(image_id, rows, cols, data) VALUES(?, ?, ?, ?)"));
    sqlite3_bind_int64(st.s, 1, image_id);
    sqlite3_bind_int64(st.s, 2, rows);
    sqlite3_bind_int64(st.s, 3, cols);
    const size_t nbytes = static_cast<size_t>(rows) * cols * sizeof(float);
    sqlite3_bind_blob64(st.s, 4, nbytes ? reinterpret_cast<const char*>(data) : nullptr, nbytes, SQLITE_STATIC);
    st.Step();
}
void Database::WriteDescriptors(image_t image_id, const uint8_t* data, uint32_t rows) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("INSERT INTO descriptors(image_id, rows, cols, data) VALUES(?, ?, ?, ?)"));
    sqlite3_bind_int64(st.s, 1, image_id);
    sqlite3_bind_int64(st.s, 2, rows);
    sqlite3_bind_int64(st.s, 3, 128);
    const size_t nbytes = static_cast<size_t>(rows) * 128;
    sqlite3_bind_blob64(st.s, 4, nbytes ? reinterpret_cast<const char*>(data) : nullptr, nbytes, SQLITE_STATIC);
    st.Step();
}

static std::vector<image_pair_t> PairIdsOf(sqlite3* db, sqlite3_stmt* s) {
    std::vector<image_pair_t> ids;
    for (;;) {
        const int rc = sqlite3_step(s);
        if (rc == SQLITE_ROW) {
            ids.push_back(static_cast<image_pair_t>(sqlite3_column_int64(s, 0)));
            continue;
        }
        sqlite3_reset(s);
        if (rc != SQLITE_DONE) throw std::runtime_error(std::string("SQLite step failed: ") + sqlite3_errmsg(db));
        return ids;
    }
}
std::vector<image_pair_t> Database::ReadMatchedPairIds() const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    return PairIdsOf(db_, Prepared("SELECT pair_id FROM matches"));
}
std::vector<image_pair_t> Database::ReadVerifiedPairIds() const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    return PairIdsOf(db_, Prepared("SELECT pair_id FROM two_view_geometries"));
}
bool Database::ExistsPair(const char* table, image_pair_t pair_id) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared((std::string("SELECT 1 FROM ") + table + " WHERE pair_id = ?")));
    sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(pair_id));
    return st.Step();
}
bool Database::ExistsMatches(image_t id1, image_t id2) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    return ExistsPair("matches", ImagePairToPairId(id1, id2));
}
bool Database::ExistsInlierMatches(image_t id1, image_t id2) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    return ExistsPair("two_view_geometries", ImagePairToPairId(id1, id2));
}

static std::vector<uint32_t> BlobToMatches(sqlite3_stmt* s, int col_rows, int col_data, bool swap) {
    const uint32_t rows = static_cast<uint32_t>(sqlite3_column_int64(s, col_rows));
    std::vector<uint32_t> m(static_cast<size_t>(rows) * 2);
    if (rows) {
        if (static_cast<size_t>(sqlite3_column_bytes(s, col_data)) != m.size() * sizeof(uint32_t))
            throw std::runtime_error("matches blob has inconsistent shape");
        std::memcpy(m.data(), sqlite3_column_blob(s, col_data), m.size() * sizeof(uint32_t));
        if (swap)
            for (size_t i = 0; i + 1 < m.size(); i += 2) std::swap(m[i], m[i + 1]);
    }
    return m;
}
std::vector<uint32_t> Database::ReadMatches(image_t id1, image_t id2) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("SELECT rows, cols, data FROM matches WHERE pair_id = ?"));
    sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(id1, id2)));
    if (!st.Step()) return {};
    return BlobToMatches(st.s, 0, 2, SwapImagePair(id1, id2));
}
TwoViewGeometryRow Database::ReadTwoViewGeometry(image_t id1, image_t id2) const {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    TwoViewGeometryRow t;
    Stmt st(db_, Prepared("SELECT rows, cols, data, config, F, E, H, qvec, tvec FROM two_view_geometries WHERE pair_id = ?"));
    sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(id1, id2)));
    if (!st.Step()) return t;
    t.inlier_matches = BlobToMatches(st.s, 0, 2, false);
    t.config = sqlite3_column_int(st.s, 3);
    auto rd = [&](int col, double* dst, int n) {
        if (sqlite3_column_bytes(st.s, col) == static_cast<int>(n * sizeof(double)))
            std::memcpy(dst, sqlite3_column_blob(st.s, col), n * sizeof(double));
    };
    rd(4, t.F.data(), 9);
    rd(5, t.E.data(), 9);
    rd(6, t.H.data(), 9);
    rd(7, t.qvec.data(), 4);
    rd(8, t.tvec.data(), 3);
    if (SwapImagePair(id1, id2)) t.Invert();
    return t;
}

void Database::WriteMatches(image_t id1, image_t id2, const std::vector<uint32_t>& matches) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    // the blob is bound in place (SQLITE_STATIC: it outlives the step); a swapped copy is made only
    // for pairs given in descending id order
    std::vector<uint32_t> swapped;
    const std::vector<uint32_t>* m = &matches;
    if (SwapImagePair(id1, id2)) {
        swapped = matches;
        for (size_t i = 0; i + 1 < swapped.size(); i += 2) std::swap(swapped[i], swapped[i + 1]);
        m = &swapped;
    }
    Stmt st(db_, Prepared("INSERT INTO matches(pair_id, rows, cols, data) VALUES(?, ?, ?, ?)"));
    sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(id1, id2)));
    sqlite3_bind_int64(st.s, 2, static_cast<sqlite3_int64>(m->size() / 2));
    sqlite3_bind_int64(st.s, 3, 2);
    sqlite3_bind_blob(st.s, 4, m->empty() ? nullptr : reinterpret_cast<const char*>(m->data()),
                      static_cast<int>(m->size() * sizeof(uint32_t)), SQLITE_STATIC);
    st.Step();
}
void Database::WriteTwoViewGeometry(image_t id1, image_t id2, const TwoViewGeometryRow& in) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    TwoViewGeometryRow inverted;
    const bool swap = SwapImagePair(id1, id2);
    if (swap) {
        inverted = in;
        inverted.Invert();
    }
    const TwoViewGeometryRow& t = swap ? inverted : in;
    Stmt st(db_, Prepared("INSERT INTO two_view_geometries(pair_id, rows, cols, data, config, F, E, H, qvec, tvec) "
                 "VALUES(?, ?, ?, ?, ?, ?, ?, ?, ?, ?)"));
    sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(id1, id2)));
    sqlite3_bind_int64(st.s, 2, static_cast<sqlite3_int64>(t.inlier_matches.size() / 2));
    sqlite3_bind_int64(st.s, 3, 2);
    sqlite3_bind_blob(st.s, 4, t.inlier_matches.empty() ? nullptr : reinterpret_cast<const char*>(t.inlier_matches.data()),
                      static_cast<int>(t.inlier_matches.size() * sizeof(uint32_t)), SQLITE_STATIC);
    sqlite3_bind_int64(st.s, 5, t.config);
    // COLMAP stores the matrices only when there are inlier matches, empty blobs otherwise
    const bool has = !t.inlier_matches.empty();
    auto wr = [&](int col, const double* src, int n) {
        sqlite3_bind_blob(st.s, col, has ? reinterpret_cast<const char*>(src) : nullptr, has ? static_cast<int>(n * sizeof(double)) : 0,
                          SQLITE_STATIC);
    };
    wr(6, t.F.data(), 9);
    wr(7, t.E.data(), 9);
    wr(8, t.H.data(), 9);
    wr(9, t.qvec.data(), 4);
    wr(10, t.tvec.data(), 3);
    st.Step();
}
void Database::DeleteMatches(image_t id1, image_t id2) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("DELETE FROM matches WHERE pair_id = ?"));
    sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(id1, id2)));
    st.Step();
}
void Database::DeleteInlierMatches(image_t id1, image_t id2) {
    std::lock_guard<std::recursive_mutex> lock(mu_);
    Stmt st(db_, Prepared("DELETE FROM two_view_geometries WHERE pair_id = ?"));
    sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(id1, id2)));
    st.Step();
}

}  // namespace amchost
