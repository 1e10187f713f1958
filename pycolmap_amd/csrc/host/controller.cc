#include "controller.h"

#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <fstream>
#include <future>
#include <memory>
#include <sstream>
#include <unordered_map>
#include <unordered_set>

namespace amchost {

namespace {
void Check(int rc, const char* what) {
    if (rc != AMC_OK) throw AmcFailure(rc, std::string(what) + ": " + amc_last_error());
}
double NowMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// ExtractInlierMatches; for a MULTIPLE geometry (mask byte = 1 + geometry index) the per-geometry
// lists follow one another, as EstimateMultipleTwoViewGeometries concatenates them
void AppendInlierMatches(const uint8_t* mask, const uint32_t* matches, size_t m, std::vector<uint32_t>* out) {
    uint8_t top = 0;
    size_t n = 0;
    for (size_t i = 0; i < m; ++i) {
        top = std::max(top, mask[i]);
        n += mask[i] != 0;
    }
    const size_t base = out->size();
    out->resize(base + 2 * n);
    uint32_t* dst = out->data() + base;
    if (top <= 1) {  // the usual single geometry: one pass, no per-element capacity checks
        for (size_t i = 0; i < m; ++i)
            if (mask[i]) {
                *dst++ = matches[2 * i];
                *dst++ = matches[2 * i + 1];
            }
        return;
    }
    for (uint8_t g = 1; g <= top; ++g)
        for (size_t i = 0; i < m; ++i)
            if (mask[i] == g) {
                *dst++ = matches[2 * i];
                *dst++ = matches[2 * i + 1];
            }
}

amc_tvg_opts ToAmc(const TwoViewGeometryOptions& o) {
    amc_tvg_opts t;
    amc_tvg_opts_default(&t);
    t.min_num_inliers = o.min_num_inliers;
    t.min_E_F_inlier_ratio = o.min_E_F_inlier_ratio;
    t.max_H_inlier_ratio = o.max_H_inlier_ratio;
    t.watermark_min_inlier_ratio = o.watermark_min_inlier_ratio;
    t.watermark_border_size = o.watermark_border_size;
    t.detect_watermark = o.detect_watermark;
    t.multiple_ignore_watermark = o.multiple_ignore_watermark;
    t.force_H_use = o.force_H_use;
    t.compute_relative_pose = o.compute_relative_pose;
    t.multiple_models = o.multiple_models;
    t.ransac.max_error = o.ransac_options.max_error;
    t.ransac.min_inlier_ratio = o.ransac_options.min_inlier_ratio;
    t.ransac.confidence = o.ransac_options.confidence;
    t.ransac.dyn_num_trials_multiplier = o.ransac_options.dyn_num_trials_multiplier;
    t.ransac.min_num_trials = static_cast<int64_t>(o.ransac_options.min_num_trials);
    t.ransac.max_num_trials = static_cast<int64_t>(std::min<size_t>(o.ransac_options.max_num_trials, size_t(1) << 30));
    return t;
}

MatchController::MatchController(const std::string& database_path, const SiftMatchingOptions& sift,
                                 const TwoViewGeometryOptions& tvg, std::vector<int> device_ids)
    : path_(database_path), sift_(sift), tvg_(tvg), device_ids_(std::move(device_ids)) {
    if (device_ids_.empty()) device_ids_.push_back(0);
}

MatchController::~MatchController() {
    for (amc_ctx* c : ctxs_)
        if (c) amc_ctx_destroy(c);
}

uint32_t MatchController::SlotOf(image_t id) const {
    if (id >= slot_of_image_.size() || slot_of_image_[id] == 0xFFFFFFFFu)
        throw std::invalid_argument("unknown image_id " + std::to_string(id));
    return slot_of_image_[id];
}

// FeatureMatcherCache::Setup + the GPU matcher's descriptor upload.  The LRU cache over SQLite is
// replaced by a device-resident arena holding every image.
void MatchController::Setup() {
    struct SetupTimer {
        double t0, *acc;
        ~SetupTimer() { *acc += NowMs() - t0; }
    } setup_timer{NowMs(), &stats.setup_ms};
    db_ = std::make_unique<Database>(path_);
    db_->SetBulkWriteMode(true);  // rollback journal while this controller appends; WAL again on close
    {
        const std::vector<image_pair_t> m = db_->ReadMatchedPairIds(), t = db_->ReadVerifiedPairIds();
        had_matches_.insert(m.begin(), m.end());
        had_tvg_.insert(t.begin(), t.end());
    }
    images_ = db_->ReadAllImages();
    const std::vector<CameraRow> cams = db_->ReadAllCameras();
    std::unordered_map<camera_t, const CameraRow*> cam_by_id;
    for (const auto& c : cams) cam_by_id[c.camera_id] = &c;
    for (int dev : device_ids_) {
        amc_ctx* c = nullptr;
        Check(amc_ctx_create(dev, &c), "amc_ctx_create");
        ctxs_.push_back(c);
        Check(amc_ctx_reserve_slots(c, static_cast<uint32_t>(images_.size())), "amc_ctx_reserve_slots");
    }
    ctx_ = ctxs_[0];
    image_t max_id = 0;
    for (const auto& im : images_) max_id = std::max(max_id, im.image_id);
    slot_of_image_.assign(static_cast<size_t>(max_id) + 1, 0xFFFFFFFFu);
    desc_rows_.assign(images_.size(), 0);
    // The arena (descriptors, keypoints, camera of every image) is replicated on every context: 0.26 - 5.2 GB for
    // BASELINE's configs against 288 GB of HBM per GPU.  One reader (SQLite), one uploader
    // thread per context working through the same rows.
    struct Row {
        std::vector<uint8_t> desc;
        std::vector<float> kp;
        uint32_t use = 0, krows = 0;
        const CameraRow* cam = nullptr;
    };
    auto upload = [&](amc_ctx* c, uint32_t s, const Row& r) {
        Check(amc_upload_descriptors(c, s, r.desc.data(), r.use), "amc_upload_descriptors");
        Check(amc_upload_keypoints(c, s, r.kp.data(), r.krows, 2), "amc_upload_keypoints");
        Check(amc_upload_camera(c, s, r.cam->model_id, r.cam->width, r.cam->height, r.cam->params.data(),
                                static_cast<int32_t>(r.cam->params.size()), r.cam->has_prior_focal_length),
              "amc_upload_camera");
    };
    constexpr uint32_t kChunk = 64;  // images read from SQLite per round of uploads
    const uint32_t nimg = static_cast<uint32_t>(images_.size());
    for (uint32_t s = 0; s < nimg; ++s) slot_of_image_[images_[s].image_id] = s;
    // SQLite is read by a worker thread one chunk ahead of the uploads (reading 295 MB of blobs and moving them to the
    // device cost about the same: one hides behind the other)
    auto read_chunk = [&](uint32_t s0) {
        const uint32_t s1 = std::min<uint32_t>(s0 + kChunk, nimg);
        std::vector<Row> rows(s1 - s0);
        for (uint32_t s = s0; s < s1; ++s) {
            const ImageRow& im = images_[s];
            Row& r = rows[s - s0];
            uint32_t drows = 0;
            r.desc = db_->ReadDescriptors(im.image_id, &drows);
            r.kp = db_->ReadKeypointsXY(im.image_id, &r.krows);
            // COLMAP's GPU matcher clamps to the first max_num_matches features
            // (WarnIfMaxNumMatchesReachedGPU; SiftMatchingOptions.max_num_matches)
            r.use = std::min<uint32_t>(drows, static_cast<uint32_t>(std::max(sift_.max_num_matches, 0)));
            auto it = cam_by_id.find(im.camera_id);
            if (it == cam_by_id.end()) throw std::runtime_error("image " + im.name + " references a missing camera");
            r.cam = it->second;
        }
        return rows;
    };
    std::future<std::vector<Row>> ahead;
    if (nimg) ahead = std::async(std::launch::async, read_chunk, 0u);
    for (uint32_t s0 = 0; s0 < nimg; s0 += kChunk) {
        const uint32_t s1 = std::min<uint32_t>(s0 + kChunk, nimg);
        const std::vector<Row> rows = ahead.get();  // (rethrows what the reader threw)
        if (s1 < nimg) ahead = std::async(std::launch::async, read_chunk, s1);
        struct Drain {  // an upload that throws must not leave the reader running into a dying controller
            std::future<std::vector<Row>>& f;
            ~Drain() {
                if (f.valid()) f.wait();
            }
        } drain{ahead};
        for (uint32_t s = s0; s < s1; ++s) desc_rows_[s] = rows[s - s0].use;
        if (ctxs_.size() == 1) {
            for (uint32_t s = s0; s < s1; ++s) upload(ctx_, s, rows[s - s0]);
        } else {
            std::vector<std::future<void>> up;
            for (amc_ctx* c : ctxs_)
                up.push_back(std::async(std::launch::async, [&, c] {
                    for (uint32_t s = s0; s < s1; ++s) upload(c, s, rows[s - s0]);
                }));
            for (auto& f : up) f.get();
        }
    }
}

// FeatureMatcherController::Match (colmap/controllers/feature_matching_utils.cc), batched
void MatchController::Match(const ImagePairs& image_pairs) {
    std::vector<Job> jobs = Compute(image_pairs);
    DatabaseTransaction tx(db_.get());
    Write(jobs);
    tx.Commit();
}

std::vector<MatchController::Job> MatchController::Compute(const ImagePairs& image_pairs) {
    std::vector<Job> jobs;
    if (image_pairs.empty()) return jobs;
    struct TotalTimer {
        double t0, *acc;
        ~TotalTimer() { *acc += NowMs() - t0; }
    } total_timer{NowMs(), &stats.match_total_ms};
    std::unordered_set<image_pair_t> seen;
    seen.reserve(image_pairs.size());
    const double t_db0 = NowMs();
    for (const auto& pr : image_pairs) {
        if (pr.first == pr.second) continue;  // avoid self-matches
        const image_pair_t pid = Database::ImagePairToPairId(pr.first, pr.second);
        if (!seen.insert(pid).second) continue;  // avoid duplicate image pairs
        if (computed_.count(pid)) { ++stats.pairs_skipped; continue; }  // done earlier in this run (rows may be in flight)
        const bool exists_matches = had_matches_.count(pid) != 0;
        const bool exists_inlier = had_tvg_.count(pid) != 0;
        if (exists_matches && exists_inlier) { ++stats.pairs_skipped; continue; }  // resume
        // One of the two rows missing: recompute from scratch.  COLMAP deletes what exists right here; the
        // deletes are deferred to Write(), into the transaction that inserts the replacement rows, so that a
        // call that fails on the way (a stored match indexing past the keypoints, an option the device
        // rejects) leaves the database as it found it.
        Job j;
        j.id1 = pr.first;
        j.id2 = pr.second;
        j.have_matches = exists_matches;
        j.had_tvg_row = exists_inlier;
        if (exists_matches) j.matches = db_->ReadMatches(pr.first, pr.second);
        jobs.push_back(std::move(j));
    }
    stats.db_ms += NowMs() - t_db0;
    if (jobs.empty()) return jobs;

    // ---- the group's pairs are dealt to the contexts in contiguous ranges of equal matching work (sum of n1 * n2;
    //      pairs with stored matches cost only their verification and weigh 0 here).  Pairs are independent and
    //      every pair re-seeds its generator, so what a range computes does not depend on which context took it or
    //      on how many there are: the rows written are the same for any gpu_index.
    const size_t D = std::min(ctxs_.size(), jobs.size());
    if (D <= 1) {
        ComputeOn(ctx_, jobs, 0, jobs.size());
    } else {
        std::vector<double> cum(jobs.size() + 1, 0.0);
        for (size_t k = 0; k < jobs.size(); ++k) {
            double w = 1.0;  // never 0: a group of stored-match pairs still splits evenly
            if (!jobs[k].have_matches)
                w += static_cast<double>(desc_rows_[SlotOf(jobs[k].id1)]) * static_cast<double>(desc_rows_[SlotOf(jobs[k].id2)]);
            cum[k + 1] = cum[k] + w;
        }
        std::vector<size_t> cut(D + 1, jobs.size());
        cut[0] = 0;
        for (size_t d = 1; d < D; ++d)
            cut[d] = static_cast<size_t>(std::lower_bound(cum.begin(), cum.end(), cum.back() * d / D) - cum.begin());
        for (size_t d = 1; d <= D; ++d) cut[d] = std::max(cut[d], cut[d - 1]);
        std::vector<std::future<void>> work;
        for (size_t d = 0; d < D; ++d)
            if (cut[d] < cut[d + 1])
                work.push_back(std::async(std::launch::async, [this, &jobs, &cut, d] { ComputeOn(ctxs_[d], jobs, cut[d], cut[d + 1]); }));
        std::exception_ptr first;
        for (auto& f : work) {
            try {
                f.get();
            } catch (...) {
                if (!first) first = std::current_exception();
            }
        }
        if (first) std::rethrow_exception(first);
    }

    // only now: a failed call above must leave these pairs eligible for the next attempt
    for (const Job& j : jobs) computed_.insert(Database::ImagePairToPairId(j.id1, j.id2));
    return jobs;
}

// FeatureMatcherWorker + VerifierWorker for jobs[begin, end) on one context
void MatchController::ComputeOn(amc_ctx* ctx, std::vector<Job>& all_jobs, size_t begin, size_t end) {
    struct Range {  // the code below indexes jobs[0 .. size())
        std::vector<Job>& v;
        size_t b, e;
        size_t size() const { return e - b; }
        Job& operator[](size_t k) const { return v[b + k]; }
        Job* begin() const { return v.data() + b; }
        Job* end() const { return v.data() + e; }
    } jobs{all_jobs, begin, end};
    MatchStats stats;  // this range's share; merged under the lock at the end
    amc_ctx* const ctx_ = ctx;
    // ---- FeatureMatcherWorker: descriptor matching for the pairs without stored matches ----
    std::vector<uint32_t> s1, s2;
    std::vector<size_t> which;
    for (size_t k = 0; k < jobs.size(); ++k)
        if (!jobs[k].have_matches) {
            s1.push_back(SlotOf(jobs[k].id1));
            s2.push_back(SlotOf(jobs[k].id2));
            which.push_back(k);
        }
    amc_match_opts mo;
    amc_match_opts_default(&mo);
    mo.max_ratio = sift_.max_ratio;
    mo.max_distance = sift_.max_distance;
    mo.cross_check = sift_.cross_check;
    const amc_tvg_opts to = ToAmc(tvg_);
    const size_t min_inl = static_cast<size_t>(std::max(tvg_.min_num_inliers, 0));
    // Nothing stored for any pair of the group (the usual match_* run): one fused call, the verification kernel
    // reads every pair's matches where the matcher left them in HBM (no packing of a second match list on the
    // host, no upload of it).  Pairs below min_num_inliers come back DEGENERATE and are dropped in Write().
    const bool fused = which.size() == jobs.size() && !tvg_.multiple_models;
    amc_verify_result vr;
    std::memset(&vr, 0, sizeof vr);
    std::vector<uint32_t> v1, v2, vmatches;
    std::vector<uint64_t> voff{0};
    std::vector<size_t> vwhich;
    bool have_vr = false;
    auto take_matches = [&](const amc_match_result& r) {
        for (size_t p = 0; p < which.size(); ++p) {
            Job& j = jobs[which[p]];
            j.matches.assign(r.matches + 2 * r.offsets[p], r.matches + 2 * r.offsets[p + 1]);
        }
        stats.match_device_ms += r.device_ms;
        stats.num_distances += r.num_distances;
        stats.pairs_matched += which.size();
    };
    if (fused) {
        amc_match_result r;
        const double t_call = NowMs();
        Check(amc_match_verify_pairs(ctx_, s1.data(), s2.data(), which.size(), &mo, &to, /*seed=*/0, &r, &vr),
              "amc_match_verify_pairs");
        stats.match_call_ms += NowMs() - t_call;  // both stages: the device times below split it
        {
            double tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (amc_ctx_last_timeline(ctx_, tl) == AMC_OK) {
                stats.fused_setup_ms += tl[0];
                stats.fused_match_ms += tl[1] - tl[0];
                stats.fused_launch_ms += tl[2] - tl[1];
                stats.fused_verify_wait_ms += tl[3] - tl[2];
                stats.fused_handover_hidden_ms += tl[5];
            }
        }
        take_matches(r);
        v1 = s1;
        v2 = s2;
        vwhich = which;
        voff.assign(r.offsets, r.offsets + which.size() + 1);
        amc_match_result_free(&r);
        have_vr = true;
        for (const Job& j : jobs) stats.pairs_verified += j.matches.size() / 2 >= min_inl;
    } else if (!which.empty()) {
        amc_match_result r;
        const double t_call = NowMs();
        Check(amc_match_pairs(ctx_, s1.data(), s2.data(), which.size(), &mo, &r), "amc_match_pairs");
        stats.match_call_ms += NowMs() - t_call;
        take_matches(r);
        amc_match_result_free(&r);
    }

    // ---- VerifierWorker: only pairs with >= min_num_inliers matches are estimated ----------
    if (!fused) {
        size_t total = 0;
        for (const Job& j : jobs)
            if (j.matches.size() / 2 >= min_inl) total += j.matches.size();
        vmatches.reserve(total);
        for (size_t k = 0; k < jobs.size(); ++k) {
            const size_t m = jobs[k].matches.size() / 2;
            if (m >= min_inl) {
                v1.push_back(SlotOf(jobs[k].id1));
                v2.push_back(SlotOf(jobs[k].id2));
                vmatches.insert(vmatches.end(), jobs[k].matches.begin(), jobs[k].matches.end());
                voff.push_back(voff.back() + m);
                vwhich.push_back(k);
            }
        }
        if (!vwhich.empty()) {
            const double t_call = NowMs();
            Check(amc_verify_pairs(ctx_, v1.data(), v2.data(), vwhich.size(), voff.data(), vmatches.data(), &to,
                                   /*seed=*/0, &vr),
                  "amc_verify_pairs");
            stats.verify_call_ms += NowMs() - t_call;
            have_vr = true;
            stats.pairs_verified += vwhich.size();
        }
    }
    if (have_vr) {
        for (size_t p = 0; p < vwhich.size(); ++p) {
            Job& j = jobs[vwhich[p]];
            if (j.matches.size() / 2 < min_inl) continue;  // fused call: not a pair COLMAP's verifier estimates
            const amc_tvg& g = vr.tvg[p];
            j.tvg.config = g.config;
            std::memcpy(j.tvg.E.data(), g.E, sizeof g.E);
            std::memcpy(j.tvg.F.data(), g.F, sizeof g.F);
            std::memcpy(j.tvg.H.data(), g.H, sizeof g.H);
            if (vr.pose) {  // compute_relative_pose: cam2_from_cam1 goes to the qvec / tvec columns
                std::memcpy(j.tvg.qvec.data(), vr.pose[p].qvec, sizeof vr.pose[p].qvec);
                std::memcpy(j.tvg.tvec.data(), vr.pose[p].tvec, sizeof vr.pose[p].tvec);
            }
            const uint8_t* mask = vr.inlier_mask + voff[p];
            const size_t m = j.matches.size() / 2;
            AppendInlierMatches(mask, j.matches.data(), m, &j.tvg.inlier_matches);
        }
        stats.verify_device_ms += vr.device_ms;

        // ---- guided matching (FeatureMatcherWorker with a verified geometry): pairs that kept at
        //      least min_num_inliers inliers and whose configuration COLMAP guides on are matched
        //      again under the geometric filter; the result REPLACES the inlier matches, the raw
        //      matches and the models stay as they are.
        if (sift_.guided_matching) {
            std::vector<uint32_t> g1, g2;
            std::vector<amc_tvg> geoms;
            std::vector<size_t> gwhich;
            for (size_t p = 0; p < vwhich.size(); ++p) {
                const Job& j = jobs[vwhich[p]];
                const int cfg = vr.tvg[p].config;
                const bool guides = cfg == AMC_TVG_CALIBRATED || cfg == AMC_TVG_UNCALIBRATED ||
                                    cfg == AMC_TVG_PLANAR || cfg == AMC_TVG_PANORAMIC ||
                                    cfg == AMC_TVG_PLANAR_OR_PANORAMIC;
                if (!guides || j.tvg.inlier_matches.size() / 2 < min_inl) continue;
                g1.push_back(v1[p]);
                g2.push_back(v2[p]);
                geoms.push_back(vr.tvg[p]);
                gwhich.push_back(vwhich[p]);
            }
            if (!gwhich.empty()) {
                amc_match_result gr;
                Check(amc_match_guided_pairs(ctx_, g1.data(), g2.data(), gwhich.size(), geoms.data(),
                                             tvg_.ransac_options.max_error, &mo, &gr),
                      "amc_match_guided_pairs");
                for (size_t q = 0; q < gwhich.size(); ++q)
                    jobs[gwhich[q]].tvg.inlier_matches.assign(gr.matches + 2 * gr.offsets[q],
                                                              gr.matches + 2 * gr.offsets[q + 1]);
                stats.guided_device_ms += gr.device_ms;
                stats.pairs_guided += gwhich.size();
                amc_match_result_free(&gr);
            }
        }
        amc_verify_result_free(&vr);
    }

    {
        std::lock_guard<std::mutex> lock(stats_mu_);
        MatchStats& t = this->stats;
        t.pairs_matched += stats.pairs_matched; t.pairs_verified += stats.pairs_verified; t.pairs_guided += stats.pairs_guided;
        t.match_device_ms += stats.match_device_ms; t.verify_device_ms += stats.verify_device_ms;
        t.guided_device_ms += stats.guided_device_ms; t.match_call_ms += stats.match_call_ms;
        t.verify_call_ms += stats.verify_call_ms; t.num_distances += stats.num_distances;
        t.fused_setup_ms += stats.fused_setup_ms; t.fused_match_ms += stats.fused_match_ms; t.fused_launch_ms += stats.fused_launch_ms;
        t.fused_verify_wait_ms += stats.fused_verify_wait_ms; t.fused_handover_hidden_ms += stats.fused_handover_hidden_ms;
    }
}

// ---- controller thread: drop results below min_num_inliers, write both tables ----------
void MatchController::Write(std::vector<Job>& jobs) {
    const size_t min_inl = static_cast<size_t>(std::max(tvg_.min_num_inliers, 0));
    const double t_db1 = NowMs();
    for (Job& j : jobs) {
        if (j.matches.size() / 2 < min_inl) j.matches.clear();
        if (j.tvg.inlier_matches.size() / 2 < min_inl) j.tvg = TwoViewGeometryRow();
        if (j.had_tvg_row) db_->DeleteInlierMatches(j.id1, j.id2);
        if (j.have_matches) db_->DeleteMatches(j.id1, j.id2);
        db_->WriteMatches(j.id1, j.id2, j.matches);
        db_->WriteTwoViewGeometry(j.id1, j.id2, j.tvg);
    }
    stats.write_ms += NowMs() - t_db1;
}

// ---- loop detection -------------------------------------------------------------------------
// COLMAP's SequentialFeatureMatcher::RunLoopDetection asks a FLANN vocabulary tree (a file the
// user has to supply) for the images most similar to every loop_detection_period-th image and
// matches those pairs.  There is no FLANN here and an approximate index is not a parity target;
// the retrieval is replaced by exact FEATURE VOTING with the kernels that already exist: every
// image keeps a copy of its first `max_features` descriptors, the query's copy is matched against
// every candidate's copy with the regular matcher (ratio test + cross check), and the candidates
// are ranked by the number of matches.  It is deterministic, needs no training data, and costs
// N x max_features^2 distances per query (6.7e8 at N = 10,000, 256 features: well under a ms of
// the scan kernel).  tests/test_pipeline_gpu.py restates it with the CPU oracle.
void MatchController::SetupLoopIndex(int max_features) {
    if (loop_index_features_ == max_features) return;
    const uint32_t n = static_cast<uint32_t>(images_.size());
    Check(amc_ctx_grow_slots(ctx_, 2 * n), "amc_ctx_grow_slots");
    for (uint32_t s = 0; s < n; ++s) {
        uint32_t drows = 0;
        const std::vector<uint8_t> desc = db_->ReadDescriptors(images_[s].image_id, &drows);
        const uint32_t use = std::min<uint32_t>({drows, static_cast<uint32_t>(max_features),
                                                 static_cast<uint32_t>(std::max(sift_.max_num_matches, 0))});
        Check(amc_upload_descriptors(ctx_, n + s, desc.data(), use), "amc_upload_descriptors (loop index)");
    }
    loop_index_features_ = max_features;
}

std::vector<image_t> MatchController::RetrieveLoopCandidates(image_t query, const std::vector<image_t>& candidates,
                                                             int num_images, int max_features) {
    return RetrieveLoopCandidatesBatch({query}, candidates, num_images, max_features)[0];
}

// Several queries in ONE device call (round 6): a query against 10^4 candidates is 10^4 pairs of max_features^2 - a
// third of a millisecond of scan behind a call's fixed costs; the pairs of a batch of queries are independent, so the
// votes, and with them the retrieved images, are those of one call per query.
std::vector<std::vector<image_t>> MatchController::RetrieveLoopCandidatesBatch(const std::vector<image_t>& queries,
                                                                               const std::vector<image_t>& candidates,
                                                                               int num_images, int max_features) {
    std::vector<std::vector<image_t>> found(queries.size());
    if (num_images <= 0 || candidates.empty() || queries.empty()) return found;
    SetupLoopIndex(max_features > 0 ? max_features : kLoopIndexDefaultFeatures);
    const uint32_t n = static_cast<uint32_t>(images_.size());
    std::vector<uint32_t> s1, s2;
    std::vector<image_t> who;
    std::vector<size_t> first(queries.size() + 1, 0);  // query q's pairs: [first[q], first[q + 1])
    s1.reserve(queries.size() * candidates.size());
    s2.reserve(queries.size() * candidates.size());
    who.reserve(queries.size() * candidates.size());
    for (size_t q = 0; q < queries.size(); ++q) {
        const uint32_t qs = n + SlotOf(queries[q]);
        for (image_t c : candidates) {
            if (c == queries[q]) continue;
            s1.push_back(qs);
            s2.push_back(n + SlotOf(c));
            who.push_back(c);
        }
        first[q + 1] = who.size();
    }
    if (who.empty()) return found;
    amc_match_opts mo;
    amc_match_opts_default(&mo);
    mo.max_ratio = sift_.max_ratio;
    mo.max_distance = sift_.max_distance;
    mo.cross_check = sift_.cross_check;
    amc_match_result r;
    Check(amc_match_pairs(ctx_, s1.data(), s2.data(), who.size(), &mo, &r), "amc_match_pairs (loop index)");
    stats.loop_device_ms += r.device_ms;
    stats.loop_pairs_scored += who.size();
    stats.loop_queries += queries.size();
    std::vector<size_t> order;
    for (size_t q = 0; q < queries.size(); ++q) {
        const size_t b = first[q], e = first[q + 1];
        order.resize(e - b);
        for (size_t i = b; i < e; ++i) order[i - b] = i;
        auto votes = [&](size_t i) { return r.offsets[i + 1] - r.offsets[i]; };
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t c) { return votes(a) > votes(c); });
        for (size_t k = 0; k < order.size() && found[q].size() < static_cast<size_t>(num_images); ++k)
            if (votes(order[k]) > 0) found[q].push_back(who[order[k]]);  // an image without a single vote is no candidate
    }
    amc_match_result_free(&r);
    return found;
}

// ExhaustiveFeatureMatcher::Run: block pairs, one transaction + Match() each
std::vector<ImagePairs> ExhaustiveBlocks(const std::vector<image_t>& ids, int block_size) {
    if (block_size <= 1) throw std::invalid_argument("block_size must be > 1");
    const size_t B = static_cast<size_t>(block_size);
    const size_t num_blocks = (ids.size() + B - 1) / B;
    std::vector<ImagePairs> out;
    for (size_t sb1 = 0; sb1 < num_blocks; ++sb1) {
        const size_t s1 = sb1 * B, e1 = std::min(ids.size(), s1 + B);
        for (size_t sb2 = 0; sb2 < num_blocks; ++sb2) {
            const size_t s2 = sb2 * B, e2 = std::min(ids.size(), s2 + B);
            ImagePairs pairs;
            for (size_t i1 = s1; i1 < e1; ++i1)
                for (size_t i2 = s2; i2 < e2; ++i2) {
                    const size_t b1 = i1 % B, b2 = i2 % B;
                    if ((i1 > i2 && b1 <= b2) || (i1 < i2 && b1 < b2)) pairs.emplace_back(ids[i1], ids[i2]);
                }
            out.push_back(std::move(pairs));
        }
    }
    return out;
}
// COLMAP runs one Match() + one transaction per generated block and keeps the device busy with a
// pool of worker threads.  Here the device is kept busy by batching instead: consecutive blocks are
// merged until a device call has kGroupPairs pairs (a sequential block is ~17 pairs, far too few for
// 1024 resident waves), then matched, verified and written in one transaction.  What ends up in the
// database is the same; only the granularity of a resumed, interrupted run changes.
constexpr size_t kGroupPairsDefault = 32768;  // the verification kernel's launch tail: 64 k pairs/s at 4 k pairs, 95 k at 64 k
static size_t GroupPairs() {  // AMC_GROUP_PAIRS: A/B hook (read once)
    static const size_t v = [] {
        const char* e = std::getenv("AMC_GROUP_PAIRS");
        const long long n = e ? std::atoll(e) : 0;
        return n > 0 ? static_cast<size_t>(n) : kGroupPairsDefault;
    }();
    return v;
}
static void RunGrouped(MatchController& c, const std::vector<ImagePairs>& blocks) {
    const size_t kGroupPairs = GroupPairs();
    // One group's rows are written by a worker thread (one transaction per group) while the device
    // matches and verifies the next group: SQLite's share of a run hides behind the kernels.
    ImagePairs group;
    std::future<void> writer;
    struct Drain {  // whatever happens, no writer outlives this call
        std::future<void>& f;
        ~Drain() {
            if (f.valid()) f.wait();
        }
    } drain{writer};
    auto flush = [&] {
        if (group.empty()) return;
        auto jobs = std::make_shared<std::vector<MatchController::Job>>(c.Compute(group));
        group.clear();
        if (writer.valid()) writer.get();  // one writer at a time; rethrows what it threw
        if (jobs->empty()) return;
        writer = std::async(std::launch::async, [&c, jobs] {
            DatabaseTransaction tx(&c.Db());
            c.Write(*jobs);
            tx.Commit();
        });
    };
    for (const ImagePairs& pairs : blocks) {
        if (c.StopRequested()) break;
        group.insert(group.end(), pairs.begin(), pairs.end());
        if (group.size() >= kGroupPairs) flush();
    }
    if (!c.StopRequested()) flush();
    if (writer.valid()) writer.get();
}

void RunExhaustive(MatchController& c, const ExhaustiveMatchingOptions& o) {
    std::vector<image_t> ids;
    for (const auto& im : c.Images()) ids.push_back(im.image_id);
    RunGrouped(c, ExhaustiveBlocks(ids, o.block_size));
}

// SequentialFeatureMatcher::RunSequentialMatching
std::vector<ImagePairs> SequentialBlocks(const std::vector<image_t>& ids, int overlap, bool quadratic_overlap) {
    if (overlap <= 0) throw std::invalid_argument("overlap must be > 0");
    std::vector<ImagePairs> out;
    for (size_t i1 = 0; i1 < ids.size(); ++i1) {
        ImagePairs pairs;
        for (int i = 0; i < overlap; ++i) {
            const size_t i2 = i1 + static_cast<size_t>(i);
            if (i2 >= ids.size()) break;
            pairs.emplace_back(ids[i1], ids[i2]);
            if (quadratic_overlap && i < 31) {
                const size_t i2q = i1 + (size_t(1) << i);
                if (i2q < ids.size()) pairs.emplace_back(ids[i1], ids[i2q]);
            }
        }
        out.push_back(std::move(pairs));
    }
    return out;
}
void RunSequential(MatchController& c, const SequentialMatchingOptions& o) {
    std::vector<ImageRow> ordered = c.Images();  // GetOrderedImageIds: by name
    std::sort(ordered.begin(), ordered.end(), [](const ImageRow& a, const ImageRow& b) { return a.name < b.name; });
    std::vector<image_t> ids;
    for (const auto& im : ordered) ids.push_back(im.image_id);
    RunGrouped(c, SequentialBlocks(ids, o.overlap, o.quadratic_overlap));
    // SequentialFeatureMatcher::RunLoopDetection: every loop_detection_period-th image (in name
    // order) is matched against its loop_detection_num_images retrieved images
    if (o.loop_detection) {
        std::vector<ImagePairs> loop_blocks;
        // queries per device call: as many as make ~2^16 pairs, at least one.  Measured on a 3,000-image database (300
        // queries, profiles/r06/ab_loop_v2.txt, _v3.txt): one query per call 213 ms of device time, 10 - 21 per call
        // 124 - 127, 42: 154 - 158, 84 and more: 220 - 276 (few, large calls pay the growth of their result buffers)
        size_t per_call = std::max<size_t>(1, (size_t(1) << 16) / std::max<size_t>(ids.size(), 1));
        if (const char* e = std::getenv("AMC_LOOP_QUERIES_PER_CALL")) per_call = static_cast<size_t>(std::max(1, std::atoi(e)));  // (A/B hook)
        std::vector<image_t> queries;
        auto flush = [&] {
            if (queries.empty()) return;
            const std::vector<std::vector<image_t>> found = c.RetrieveLoopCandidatesBatch(
                queries, ids, o.loop_detection_num_images, o.loop_detection_max_num_features);
            for (size_t q = 0; q < queries.size(); ++q) {
                ImagePairs pairs;
                for (image_t j : found[q]) pairs.emplace_back(queries[q], j);
                loop_blocks.push_back(std::move(pairs));
            }
            queries.clear();
        };
        for (size_t i = 0; i < ids.size() && !c.StopRequested(); i += kLoopDetectionPeriod) {
            queries.push_back(ids[i]);
            if (queries.size() >= per_call) flush();
        }
        if (!c.StopRequested()) flush();
        RunGrouped(c, loop_blocks);
    }
}

// ---- spatial matching (COLMAP 3.9.1 feature/matching.cc, SpatialFeatureMatcher::Run) ------------------------------
// COLMAP: images whose location prior is "unset" (x = y = 0 when ignore_z, x = y = z = 0 otherwise) are left out; the
// others become rows of a float [n][3] matrix (GPS priors through GPSTransform::EllToXYZ, z forced to 0 under
// ignore_z BEFORE the transform, i.e. altitude 0), a flann::LinearIndex (exhaustive, flann::L2<float>) answers
// knn = min(max_num_neighbors, n) neighbours per row, and image i is matched with its neighbours in order of
// distance until one is farther than max_distance (compared squared, as float); the query itself is skipped.
// Restated here: the same float arithmetic (L2<float> adds diff * diff left to right for a 3-vector), the same result
// order (KNNSimpleResultSet: ascending distance, equal distances in insertion = row order).
// Deviation: an image whose prior columns are NULL reads as NaN in COLMAP, passes the "unset" test and enters the
// index with NaN coordinates (its neighbours are then whatever NaN comparisons leave); here it is left out.
std::array<double, 3> EllToXYZ(const std::array<double, 3>& ell) {
    // GPSTransform's WGS84 constants: a = 6378137, b = 6356752.314245, e^2 = (a^2 - b^2) / a^2
    const double a = 6378137.0, b = 6356752.314245;
    const double e2 = (a * a - b * b) / (a * a);
    const double deg = 0.0174532925199432954743716805978692718781530857086181640625;  // DegToRad's factor
    const double lat = ell[0] * deg, lon = ell[1] * deg, alt = ell[2];
    const double sin_lat = std::sin(lat), sin_lon = std::sin(lon), cos_lat = std::cos(lat), cos_lon = std::cos(lon);
    const double N = a / std::sqrt(1 - e2 * sin_lat * sin_lat);
    return {{(N + alt) * cos_lat * cos_lon, (N + alt) * cos_lat * sin_lon, (N * (1 - e2) + alt) * sin_lat}};
}
std::vector<ImagePairs> SpatialBlocks(const std::vector<image_t>& ids, const std::vector<std::array<double, 3>>& priors,
                                      const SpatialMatchingOptions& o) {
    if (ids.size() != priors.size()) throw std::invalid_argument("ids and priors differ in length");
    if (o.max_num_neighbors <= 0) throw std::invalid_argument("max_num_neighbors must be > 0");
    if (!(o.max_distance > 0.0)) throw std::invalid_argument("max_distance must be > 0");
    std::vector<size_t> location_idxs;
    std::vector<std::array<float, 3>> loc;
    for (size_t i = 0; i < ids.size(); ++i) {
        const auto& t = priors[i];
        if (std::isnan(t[0]) || std::isnan(t[1]) || (!o.ignore_z && std::isnan(t[2]))) continue;  // (deviation above)
        if ((t[0] == 0 && t[1] == 0 && o.ignore_z) || (t[0] == 0 && t[1] == 0 && t[2] == 0 && !o.ignore_z)) continue;
        std::array<double, 3> x{{t[0], t[1], o.ignore_z ? 0.0 : t[2]}};
        if (o.is_gps) x = EllToXYZ(x);
        location_idxs.push_back(i);
        loc.push_back({{static_cast<float>(x[0]), static_cast<float>(x[1]), static_cast<float>(x[2])}});
    }
    std::vector<ImagePairs> out;
    const size_t n = loc.size();
    if (n == 0) return out;
    const size_t knn = std::min<size_t>(static_cast<size_t>(o.max_num_neighbors), n);
    const float max_distance = static_cast<float>(o.max_distance * o.max_distance);
    std::vector<std::pair<float, size_t>> best;  // the result set: ascending, at most knn entries
    for (size_t i = 0; i < n; ++i) {
        best.clear();
        for (size_t j = 0; j < n; ++j) {
            float d = 0.0f;
            for (int k = 0; k < 3; ++k) {
                const float diff = loc[i][k] - loc[j][k];
                d += diff * diff;
            }
            if (best.size() == knn && !(d < best.back().first)) continue;  // addPoint: dist >= worst is dropped
            if (best.size() < knn) best.emplace_back();
            size_t at = best.size() - 1;
            for (; at > 0 && best[at - 1].first > d; --at) best[at] = best[at - 1];
            best[at] = {d, j};
        }
        ImagePairs pairs;
        for (const auto& [d, j] : best) {
            if (j == i) continue;
            if (d > max_distance) break;
            pairs.emplace_back(ids[location_idxs[i]], ids[location_idxs[j]]);
        }
        out.push_back(std::move(pairs));
    }
    return out;
}
void RunSpatial(MatchController& c, const SpatialMatchingOptions& o) {
    std::vector<image_t> ids;
    std::vector<std::array<double, 3>> priors;
    for (const auto& im : c.Images()) {  // image_id order (COLMAP: the cache's hash-map order, which only decides ties)
        ids.push_back(im.image_id);
        priors.push_back(im.prior_t);
    }
    // (i, j) and (j, i) both appear when two images are each other's neighbours: Compute() drops the pair it has
    // seen, as COLMAP's matcher drops the one whose rows exist
    RunGrouped(c, SpatialBlocks(ids, priors, o));
}

// ImagePairsFeatureMatcher::Run: "name1 name2" lines, blank lines and '#' comments skipped
void RunImagePairs(MatchController& c, const std::string& pairs_path, int block_size) {
    std::unordered_map<std::string, image_t> by_name;
    for (const auto& im : c.Images()) by_name[im.name] = im.image_id;
    std::ifstream f(pairs_path);
    if (!f) throw std::invalid_argument("cannot read " + pairs_path);
    ImagePairs all;
    std::unordered_set<image_pair_t> seen;
    std::string line;
    while (std::getline(f, line)) {
        const size_t a = line.find_first_not_of(" \t\r\n");
        if (a == std::string::npos || line[a] == '#') continue;
        std::istringstream ss(line);
        std::string n1, n2;
        ss >> n1 >> n2;
        auto i1 = by_name.find(n1), i2 = by_name.find(n2);
        if (i1 == by_name.end() || i2 == by_name.end()) continue;  // COLMAP logs an error and skips
        const image_pair_t pid = Database::ImagePairToPairId(i1->second, i2->second);
        if (!seen.insert(pid).second) continue;
        all.emplace_back(i1->second, i2->second);
    }
    const size_t B = static_cast<size_t>(std::max(block_size, 1));
    std::vector<ImagePairs> blocks;
    for (size_t s = 0; s < all.size(); s += B)
        blocks.emplace_back(all.begin() + s, all.begin() + std::min(all.size(), s + B));
    RunGrouped(c, blocks);
}

}  // namespace amchost
