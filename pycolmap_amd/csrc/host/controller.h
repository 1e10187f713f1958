// controller.h — host scheduler of the match + verify path: what COLMAP's
// ExhaustiveFeatureMatcher / SequentialFeatureMatcher / ImagePairsFeatureMatcher +
// FeatureMatcherController + workers do, driving libamc.so.
#pragma once

#include <array>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <unordered_set>
#include <vector>

#include "../../../include/amc.h"
#include "database.h"

namespace amchost {

// option structs: field names and defaults of the pycolmap-visible classes
// (/root/reference/pycolmap/pipeline/match_features.h:73-152,
//  /root/reference/pycolmap/estimators/two_view_geometry.h:41-63)
struct SiftMatchingOptions {
    int num_threads = -1;
    std::string gpu_index = "-1";
    double max_ratio = 0.8;
    double max_distance = 0.7;
    bool cross_check = true;
    int max_num_matches = 32768;
    bool guided_matching = false;
};
struct ExhaustiveMatchingOptions {
    int block_size = 50;
};
struct SequentialMatchingOptions {
    int overlap = 10;
    bool quadratic_overlap = true;
    bool loop_detection = false;
    int loop_detection_num_images = 50;
    int loop_detection_num_nearest_neighbors = 1;
    int loop_detection_num_checks = 256;
    int loop_detection_num_images_after_verification = 0;
    int loop_detection_max_num_features = -1;
    std::string vocab_tree_path = "";
};
struct SpatialMatchingOptions {  // /root/reference/pycolmap/pipeline/match_features.h:154-175; defaults of COLMAP 3.9.1
    bool is_gps = true;
    bool ignore_z = true;
    int max_num_neighbors = 50;
    double max_distance = 100.0;
};
struct RANSACOptions {  // C++ defaults of TwoViewGeometryOptions::ransac_options
    double max_error = 4.0;
    double min_inlier_ratio = 0.25;
    double confidence = 0.999;
    double dyn_num_trials_multiplier = 3.0;
    size_t min_num_trials = 100;
    size_t max_num_trials = 10000;
};
struct TwoViewGeometryOptions {
    int min_num_inliers = 15;
    double min_E_F_inlier_ratio = 0.95;
    double max_H_inlier_ratio = 0.8;
    double watermark_min_inlier_ratio = 0.7;
    double watermark_border_size = 0.1;
    bool detect_watermark = true;
    bool multiple_ignore_watermark = true;
    bool force_H_use = false;
    bool compute_relative_pose = false;
    bool multiple_models = false;
    RANSACOptions ransac_options;
};
amc_tvg_opts ToAmc(const TwoViewGeometryOptions& o);
void AppendInlierMatches(const uint8_t* mask, const uint32_t* matches, size_t m, std::vector<uint32_t>* out);

using ImagePairs = std::vector<std::pair<image_t, image_t>>;

// exception carrying an amc status, so the bindings can map AMC_E_INVALID to ValueError
struct AmcFailure : std::runtime_error {
    int code;
    AmcFailure(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct MatchStats {
    size_t pairs_matched = 0, pairs_verified = 0, pairs_skipped = 0, pairs_guided = 0, loop_queries = 0,
           loop_pairs_scored = 0;
    double match_device_ms = 0, verify_device_ms = 0, guided_device_ms = 0, loop_device_ms = 0, db_ms = 0;
    // wall time of the library calls (device time + the library's host side) and of Match() as a whole
    double match_call_ms = 0, verify_call_ms = 0, match_total_ms = 0, setup_ms = 0;
    double write_ms = 0;  // writing both tables (a worker thread: overlaps the next group's device work)
    // amc_ctx_last_timeline of the fused calls, summed (ms): where amc_match_verify_pairs' wall time goes on the host -
    // verification set-up, the match call, closing + launching the verification slice, waiting for / packing /
    // downloading its results, and the per-batch hand-over that runs hidden beside the scans
    double fused_setup_ms = 0, fused_match_ms = 0, fused_launch_ms = 0, fused_verify_wait_ms = 0, fused_handover_hidden_ms = 0;
    uint64_t num_distances = 0;
};

class MatchController {
  public:
    // device_ids: one amc_ctx (with the whole descriptor arena) and one host thread per entry.
    // SiftMatchingOptions.gpu_index "0,1,2,3" / "-1" = all (/root/reference/pycolmap/pipeline/match_features.h:76-81,
    // where COLMAP starts one SiftGPU matcher thread per listed index); an index may be listed twice (two
    // contexts on one device), which is how the single-GPU tests cover the multi-context path.
    MatchController(const std::string& database_path, const SiftMatchingOptions& sift,
                    const TwoViewGeometryOptions& tvg, std::vector<int> device_ids);
    ~MatchController();
    void Setup();  // read cameras/images/keypoints/descriptors, fill the GPU arena
    // FeatureMatcherController::Match: filter, match, verify, write.  Match() = Compute() + Write();
    // the grouped runners call the halves themselves so that one group's rows are written (a worker
    // thread) while the device already works on the next group.
    struct Job {
        image_t id1, id2;
        bool have_matches;         // the matches row existed (its matches are verified, not recomputed)
        bool had_tvg_row = false;  // a two_view_geometries row existed (without a matches row): replaced in Write()
        std::vector<uint32_t> matches;
        TwoViewGeometryRow tvg;
    };
    void Match(const ImagePairs& pairs);
    std::vector<Job> Compute(const ImagePairs& pairs);  // filter against the DB, match, verify (all devices)
    void Write(std::vector<Job>& jobs);                  // drop what is below min_num_inliers, write both tables
    // Loop-closure candidates of `query` among `candidates` (SequentialFeatureMatcher::RunLoopDetection
    // with the vocabulary-tree query replaced by feature voting, see controller.cc): the up to
    // num_images images with the most cross-checked matches between the first max_features
    // descriptors of both images, most first, ties in `candidates` order.
    std::vector<image_t> RetrieveLoopCandidates(image_t query, const std::vector<image_t>& candidates,
                                                int num_images, int max_features);
    // the same for several queries in one device call (one list per query)
    std::vector<std::vector<image_t>> RetrieveLoopCandidatesBatch(const std::vector<image_t>& queries,
                                                                  const std::vector<image_t>& candidates, int num_images,
                                                                  int max_features);
    const std::vector<ImageRow>& Images() const { return images_; }
    Database& Db() { return *db_; }
    void RequestStop() { stop_.store(true); }
    bool StopRequested() const { return stop_.load(); }
    MatchStats stats;

  private:
    std::string path_;
    SiftMatchingOptions sift_;
    TwoViewGeometryOptions tvg_;
    std::vector<int> device_ids_;
    std::unique_ptr<Database> db_;
    std::vector<ImageRow> images_;
    std::vector<uint32_t> slot_of_image_;  // image_id -> slot (dense table)
    std::vector<uint32_t> desc_rows_;      // descriptors uploaded per slot (the weights of the work split)
    std::vector<amc_ctx*> ctxs_;           // one per entry of device_ids_; ctxs_[0] also serves the loop index
    amc_ctx* ctx_ = nullptr;               // = ctxs_[0]
    std::mutex stats_mu_;                  // the device threads add their shares to `stats`
    std::atomic<bool> stop_{false};
    // match + verify (+ guided matching) of jobs[begin, end) on one context; called concurrently for disjoint ranges
    void ComputeOn(amc_ctx* ctx, std::vector<Job>& jobs, size_t begin, size_t end);
    uint32_t SlotOf(image_t id) const;
    // pairs this run has computed already: their rows may still be on their way to the database
    std::unordered_set<image_pair_t> computed_;
    // pair ids with a row in matches / two_view_geometries when the run started (read once in Setup: the
    // resume filter then needs no query per pair and does not contend with the writer thread)
    std::unordered_set<image_pair_t> had_matches_, had_tvg_;
    int loop_index_features_ = 0;  // > 0 once the truncated copies (slots N .. 2N-1) are uploaded
    void SetupLoopIndex(int max_features);
};

constexpr int kLoopDetectionPeriod = 10;      // SequentialMatchingOptions::loop_detection_period (not bound by pycolmap)
constexpr int kLoopIndexDefaultFeatures = 512;  // features per image used for voting when max_num_features <= 0

// pair generators as pure functions: one entry per Match() call / DB transaction
std::vector<ImagePairs> ExhaustiveBlocks(const std::vector<image_t>& ids, int block_size);
std::vector<ImagePairs> SequentialBlocks(const std::vector<image_t>& ordered_ids, int overlap,
                                         bool quadratic_overlap);

// SpatialFeatureMatcher::Run's pair generation: one block per image with a location prior, its up to
// max_num_neighbors nearest images (exact search, squared float distances) closer than max_distance.
// priors[i] = Image::TvecPrior of ids[i] (NaN = NULL column).
std::vector<ImagePairs> SpatialBlocks(const std::vector<image_t>& ids, const std::vector<std::array<double, 3>>& priors,
                                      const SpatialMatchingOptions& o);
// GPSTransform(WGS84)::EllToXYZ: latitude / longitude in degrees, altitude in metres -> ECEF metres
std::array<double, 3> EllToXYZ(const std::array<double, 3>& lat_lon_alt);

void RunExhaustive(MatchController& c, const ExhaustiveMatchingOptions& o);
void RunSpatial(MatchController& c, const SpatialMatchingOptions& o);
void RunSequential(MatchController& c, const SequentialMatchingOptions& o);
void RunImagePairs(MatchController& c, const std::string& pairs_path, int block_size = 1225);

}  // namespace amchost
