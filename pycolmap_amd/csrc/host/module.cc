// module.cc — the Python-visible host layer: a C++/pybind11 extension with pycolmap's API
// surface for the match + verify path, implemented over libamc.so (C ABI) + SQLite.
//
// Mirrors (names, argument meaning, error behaviour):
//   match_exhaustive / match_sequential / match_spatial / verify_matches    /root/reference/pycolmap/pipeline/match_features.h:22-68, 219-260
//   SiftMatchingOptions / ExhaustiveMatchingOptions / SequentialMatchingOptions      ...:71-152
//   TwoViewGeometryOptions / TwoViewGeometryConfiguration / TwoViewGeometry
//                                                           /root/reference/pycolmap/estimators/two_view_geometry.h:41-93
//   RANSACOptions (Python-side defaults)                    /root/reference/pycolmap/optim/bindings.h:10-25
//   Device enum + GPU parameter check                       /root/reference/pycolmap/utils.h:9-31, main.cc:102-106
//   option "dataclass" protocol (summary/todict/mergedict, dict/kwargs ctors, implicit dict
//   conversion, copy, pickle)                               /root/reference/pycolmap/helpers.h:217-283
//   interruptible blocking wait                             /root/reference/pycolmap/helpers.h:306-347
//   Database                                                /root/reference/pycolmap/scene/database.h:9-46
//   Camera, *_matrix_estimation, estimate_two_view_geometry, squared_sampson_error -> estimators.h
#include <pybind11/numpy.h>
#include <cctype>

#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <iostream>
#include <ctime>
#include <cstdio>
#include <mutex>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <fstream>
#include <sstream>
#include <thread>

#include "controller.h"
#include "py_types.h"
#include "estimators.h"

namespace py = pybind11;
using namespace pybind11::literals;
using namespace amchost;

namespace {

enum class Device { AUTO = -1, CPU = 0, CUDA = 1 };

std::string PathToString(const py::object& p) {
    return py::module_::import("os").attr("fspath")(p).cast<std::string>();
}
// THROW_CHECK_FILE_EXISTS (/root/reference/pycolmap/log_exceptions.h:54-76, 123-125): ValueError,
// "[file:line] Check Failed: ExistsFile(path) : File <path> does not exist."
#define AMC_THROW_CHECK_FILE_EXISTS(path)                                                                        \
    do {                                                                                                         \
        std::ifstream f_(path);                                                                                  \
        if (!f_.good())                                                                                          \
            throw py::value_error(CheckMessage(__FILE__, __LINE__, "ExistsFile(" #path ")",                      \
                                               std::string("File ") + (path) + " does not exist."));             \
    } while (0)
// VerifyGPUParams analogue.  This package is the accelerator path only.
void RequireAccelerator(Device d) {
    if (d == Device::CPU)
        throw py::value_error(
            "pycolmap_amd implements the accelerated (MI355X) matcher only and has no CPU fallback; "
            "set device='auto' or device='cuda', or use the reference pycolmap for device='cpu'.");
}

// ---- option "dataclass" protocol --------------------------------------------------------------
void MergeDict(py::object self, const py::dict& d, const std::vector<std::string>& fields) {
    for (auto item : d) {
        const std::string key = py::str(item.first);
        if (std::find(fields.begin(), fields.end(), key) == fields.end()) {
            std::string known;
            for (const auto& f : fields) known += (known.empty() ? "" : ", ") + f;
            throw py::value_error(py::str(self.attr("__class__").attr("__name__")).cast<std::string>() +
                                  ": unknown option '" + key + "' (valid: " + known + ")");
        }
        py::object cur = self.attr(key.c_str());
        py::object val = py::reinterpret_borrow<py::object>(item.second);
        if (py::isinstance<py::dict>(val) && py::hasattr(cur, "mergedict")) {
            cur.attr("mergedict")(val);  // nested options: recursive merge into the defaults
            self.attr(key.c_str()) = cur;
        } else {
            self.attr(key.c_str()) = val;
        }
    }
}
py::dict ToDict(const py::object& self, const std::vector<std::string>& fields, bool recursive = true) {
    py::dict d;
    for (const auto& f : fields) {
        py::object v = self.attr(f.c_str());
        d[py::str(f)] = (recursive && py::hasattr(v, "todict")) ? v.attr("todict")() : v;
    }
    return d;
}
// CreateSummary (/root/reference/pycolmap/helpers.h:159-214): "Name:" then one line per attribute,
// "    attr = value" ("    attr: type = value" with write_type), nested option objects as "    attr: <their summary>"
// indented by four more spaces; long sequences abbreviated like the reference does.
std::string Summary(const py::object& self, const std::vector<std::string>& fields, bool write_type) {
    std::ostringstream ss;
    const std::string prefix = "    ";
    ss << py::str(self.attr("__class__").attr("__name__")).cast<std::string>() << ":";
    for (const auto& f : fields) {
        py::object v = self.attr(f.c_str());
        ss << "\n" << prefix << f;
        if (py::hasattr(v, "summary")) {
            std::string sub = v.attr("summary")(write_type).cast<std::string>();
            std::string ind;
            for (char ch : sub) {
                ind.push_back(ch);
                if (ch == '\n') ind += prefix;
            }
            ss << ": " << ind;
        } else {
            if (write_type) ss << ": " << py::str(py::type::of(v).attr("__name__")).cast<std::string>();
            std::string value = py::str(v).cast<std::string>();
            if (value.size() > 80 && py::hasattr(v, "__len__")) {
                const int n = v.attr("__len__")().cast<int>();
                value = std::string(1, value.front()) + " ... " + std::to_string(n) + " elements ... " + std::string(1, value.back());
            }
            ss << " = " << value;
        }
    }
    return ss.str();
}
// AddDefaultsToDocstrings (/root/reference/pycolmap/helpers.h:217-240): every option's docstring ends in
// "(type, default: value)", taken from a default-constructed instance
void AddDefaultsToDocstrings(const py::object& cls, const std::vector<std::string>& fields) {
    py::object obj = cls();
    for (const auto& f : fields) {
        py::object member = obj.attr(f.c_str());
        py::object prop = cls.attr(f.c_str());
        const std::string doc = py::str(prop.attr("__doc__")).cast<std::string>();
        const std::string type_name = py::str(py::type::of(member).attr("__name__")).cast<std::string>();
        std::string def = py::str(member).cast<std::string>();
        if (py::hasattr(member, "summary")) def = py::str(member.attr("__class__").attr("__name__")).cast<std::string>() + "()";
        try {
            prop.attr("__doc__") = py::str((doc == "None" ? std::string() : doc + " ") + "(" + type_name + ", default: " + def + ")");
        } catch (const py::error_already_set&) {
            PyErr_Clear();  // a read-only docstring: leave it
        }
    }
}

template <typename T>
void MakeDataclass(py::class_<T>& cls, const std::vector<std::string>& fields) {
    // dict / kwargs construction starts from the class's *Python-side* default constructor (for
    // RANSACOptions that is pycolmap's defaults, not the C++ struct's), then merges
    // (/root/reference/pycolmap/helpers.h:258-268)
    const py::object cls_obj = cls;
    cls.def(py::init([fields, cls_obj](const py::dict& d) {
        py::object self = cls_obj();
        MergeDict(self, d, fields);
        return self.cast<T>();
    }));
    cls.def(py::init([fields, cls_obj](const py::kwargs& kw) {
        py::object self = cls_obj();
        MergeDict(self, py::dict(kw), fields);
        return self.cast<T>();
    }));
    py::implicitly_convertible<py::dict, T>();
    py::implicitly_convertible<py::kwargs, T>();
    AddDefaultsToDocstrings(cls_obj, fields);
    cls.def("mergedict", [fields](py::object self, const py::dict& d) { MergeDict(self, d, fields); });
    cls.def("todict", [fields](py::object self, bool recursive) { return ToDict(self, fields, recursive); },
            "recursive"_a = true);
    cls.def("summary", [fields](py::object self, bool write_type) { return Summary(self, fields, write_type); },
            "write_type"_a = false);
    cls.def("__repr__", [fields](py::object self) { return Summary(self, fields, false); });
    cls.def("__copy__", [](const T& self) { return T(self); });
    cls.def("__deepcopy__", [](const T& self, const py::dict&) { return T(self); });
    cls.def(py::pickle([fields](py::object self) { return ToDict(self, fields, /*recursive=*/false); },
                       [fields, cls_obj](const py::dict& d) {
                           py::object self = cls_obj();
                           MergeDict(self, d, fields);
                           return self.cast<T>();
                       }));
}

// ---- interruptible blocking run (PyWait analogue) ------------------------------------------------
void RunInterruptible(MatchController& ctrl, const std::function<void()>& work) {
    std::exception_ptr err;
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    std::thread th([&] {
        try {
            work();
        } catch (...) {
            err = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> lock(mu);
            done = true;
        }
        cv.notify_one();
    });
    bool interrupted = false;
    {
        py::gil_scoped_release release;
        for (;;) {
            {   // woken when the work is done; every 50 ms to look for a pending signal
                std::unique_lock<std::mutex> lock(mu);
                if (cv.wait_for(lock, std::chrono::milliseconds(50), [&] { return done; })) break;
            }
            py::gil_scoped_acquire acquire;
            if (PyErr_CheckSignals() != 0) {  // Ctrl-C: stop cooperatively between blocks
                interrupted = true;
                ctrl.RequestStop();
                break;
            }
        }
        th.join();
    }
    if (interrupted) throw py::error_already_set();
    if (err) {
        try {
            std::rethrow_exception(err);
        } catch (const AmcFailure& e) {
            if (e.code == AMC_E_INVALID) throw py::value_error(e.what());
            throw std::runtime_error(e.what());
        } catch (const std::invalid_argument& e) {
            throw py::value_error(e.what());
        }
    }
}

py::dict StatsDict(const MatchStats& s) {
    return py::dict("pairs_matched"_a = s.pairs_matched, "pairs_verified"_a = s.pairs_verified,
                    "pairs_skipped"_a = s.pairs_skipped, "match_device_ms"_a = s.match_device_ms,
                    "verify_device_ms"_a = s.verify_device_ms, "db_ms"_a = s.db_ms,
                    "match_call_ms"_a = s.match_call_ms, "verify_call_ms"_a = s.verify_call_ms,
                    "match_total_ms"_a = s.match_total_ms, "setup_ms"_a = s.setup_ms, "write_ms"_a = s.write_ms,
                    "num_distances"_a = s.num_distances, "pairs_guided"_a = s.pairs_guided,
                    "guided_device_ms"_a = s.guided_device_ms, "loop_queries"_a = s.loop_queries,
                    "loop_pairs_scored"_a = s.loop_pairs_scored, "loop_device_ms"_a = s.loop_device_ms,
                    "fused_call_timeline_ms"_a = py::dict("verify_setup"_a = s.fused_setup_ms, "match_call"_a = s.fused_match_ms,
                                                        "close_and_launch"_a = s.fused_launch_ms,
                                                        "verify_wait_pack_download"_a = s.fused_verify_wait_ms,
                                                        "batch_handover_hidden"_a = s.fused_handover_hidden_ms));
}

}  // namespace

// SiftMatchingOptions.gpu_index (/root/reference/pycolmap/pipeline/match_features.h:76-81; COLMAP:
// "Index of the GPU used for feature matching. For multi-GPU matching, you should separate multiple GPU indices
// by comma, e.g., '0,1,2,3'", "-1" = every device): the list of devices the controller opens a context on.
// An index may repeat.  Anything that is not a list of device numbers raises instead of being ignored.
static std::vector<int> ParseGpuIndex(const std::string& gpu_index) {
    std::vector<int> out;
    std::string tok;
    auto flush = [&] {
        size_t a = 0, b = tok.size();
        while (a < b && std::isspace(static_cast<unsigned char>(tok[a]))) ++a;
        while (b > a && std::isspace(static_cast<unsigned char>(tok[b - 1]))) --b;
        const std::string t = tok.substr(a, b - a);
        tok.clear();
        if (t.empty()) throw py::value_error("gpu_index: empty entry in '" + gpu_index + "'");
        size_t pos = 0;
        int v = 0;
        try {
            v = std::stoi(t, &pos);
        } catch (const std::exception&) {
            pos = 0;
        }
        if (pos != t.size()) throw py::value_error("gpu_index: '" + t + "' is not a device number (in '" + gpu_index + "')");
        out.push_back(v);
    };
    for (char ch : gpu_index) {
        if (ch == ',') flush();
        else tok.push_back(ch);
    }
    flush();
    if (out.size() == 1 && out[0] == -1) {  // all devices
        const int n = amc_device_count();
        if (n <= 0) throw std::runtime_error(std::string("gpu_index -1: no MI355X device visible: ") + (n < 0 ? amc_last_error() : ""));
        out.clear();
        for (int i = 0; i < n; ++i) out.push_back(i);
        return out;
    }
    for (int v : out)
        if (v < 0) throw py::value_error("gpu_index: negative device number in '" + gpu_index + "' (-1 alone means all devices)");
    return out;
}

// ---- pycolmap.logging (/root/reference/pycolmap/main.cc:39-89): the glog front end the reference exposes -
// flags, per-severity destinations, info / warning / error / fatal stamped with the Python call site.  There is no
// glog here; the few lines it amounts to for this surface are written out: glog's line format, severity filtering
// by minloglevel / stderrthreshold, optional files.  fatal raises (glog aborts the process).
struct Logging {
    enum Level { INFO = 0, WARNING = 1, ERROR = 2, FATAL = 3 };
    static int minloglevel, stderrthreshold;
    static std::string log_dir;
    static bool logtostderr, alsologtostderr;
    static std::string destination[4];
    static std::mutex mu;
    static void Write(Level lv, const std::string& where, int line, const std::string& msg) {
        if (static_cast<int>(lv) < minloglevel) return;
        const auto now = std::chrono::system_clock::now();
        const std::time_t t = std::chrono::system_clock::to_time_t(now);
        const long us = static_cast<long>(std::chrono::duration_cast<std::chrono::microseconds>(now.time_since_epoch()).count() % 1000000);
        std::tm tmv;
        localtime_r(&t, &tmv);
        char head[64];
        std::snprintf(head, sizeof head, "%c%04d%02d%02d %02d:%02d:%02d.%06ld", "IWEF"[lv], tmv.tm_year + 1900, tmv.tm_mon + 1,
                      tmv.tm_mday, tmv.tm_hour, tmv.tm_min, tmv.tm_sec, us);
        std::ostringstream ln;
        ln << head << " " << std::this_thread::get_id() << " " << where << ":" << line << "] " << msg << "\n";
        std::lock_guard<std::mutex> lock(mu);
        if (logtostderr || alsologtostderr || static_cast<int>(lv) >= stderrthreshold) std::cerr << ln.str() << std::flush;
        if (!logtostderr)
            for (int k = 0; k <= static_cast<int>(lv); ++k) {  // glog: a message goes to its severity's file and all lower ones
                std::string path = destination[k];
                if (path.empty() && !log_dir.empty()) path = log_dir + "/pycolmap_amd." + "IWEF"[k] + ".log";
                if (path.empty()) continue;
                std::ofstream f(path, std::ios::app);
                f << ln.str();
            }
    }
};
int Logging::minloglevel = 0;
int Logging::stderrthreshold = 2;
std::string Logging::log_dir;
bool Logging::logtostderr = false;
bool Logging::alsologtostderr = true;   // the reference sets FLAGS_alsologtostderr = true at import
std::string Logging::destination[4];
std::mutex Logging::mu;

static std::pair<std::string, int> PythonCallFrame() {
    const py::object frame = py::module_::import("sys").attr("_getframe")(0);
    const std::string file = py::str(frame.attr("f_code").attr("co_filename"));
    const std::string function = py::str(frame.attr("f_code").attr("co_name"));
    return {file + ":" + function, py::int_(frame.attr("f_lineno"))};
}

static void BindLogging(py::module_& m) {
    py::class_<Logging> PyLogging(m, "logging");
    PyLogging.def_readwrite_static("minloglevel", &Logging::minloglevel)
        .def_readwrite_static("stderrthreshold", &Logging::stderrthreshold)
        .def_readwrite_static("log_dir", &Logging::log_dir)
        .def_readwrite_static("logtostderr", &Logging::logtostderr)
        .def_readwrite_static("alsologtostderr", &Logging::alsologtostderr)
        .def_static("set_log_destination",
                    [](Logging::Level severity, const std::string& path) { Logging::destination[severity] = path; })
        .def_static("info", [](const std::string& msg) { auto f = PythonCallFrame(); Logging::Write(Logging::INFO, f.first, f.second, msg); })
        .def_static("warning", [](const std::string& msg) { auto f = PythonCallFrame(); Logging::Write(Logging::WARNING, f.first, f.second, msg); })
        .def_static("error", [](const std::string& msg) { auto f = PythonCallFrame(); Logging::Write(Logging::ERROR, f.first, f.second, msg); })
        .def_static("fatal", [](const std::string& msg) {
            auto f = PythonCallFrame();
            Logging::Write(Logging::FATAL, f.first, f.second, msg);
            throw std::runtime_error("pycolmap.logging.fatal: " + msg);
        });
    py::enum_<Logging::Level>(PyLogging, "Level")
        .value("INFO", Logging::INFO)
        .value("WARNING", Logging::WARNING)
        .value("ERROR", Logging::ERROR)
        .value("FATAL", Logging::FATAL)
        .export_values();
}

PYBIND11_MODULE(_pycolmap, m) {
    m.doc() = "MI355X-native match + verify path behind the pycolmap API (pycolmap_amd)";
    m.attr("has_cuda") = true;  // drop-in: "an accelerator is available" (it is an MI355X)
    m.attr("has_hip") = true;
    m.attr("COLMAP_version") = "3.9.1-semantics";
    m.attr("COLMAP_build") = "pycolmap_amd (libamc.so, gfx950)";
    BindLogging(m);

    py::enum_<Device> PyDevice(m, "Device");
    PyDevice.value("auto", Device::AUTO).value("cpu", Device::CPU).value("cuda", Device::CUDA);
    PyDevice.def(py::init([](const std::string& s) {
        if (s == "auto") return Device::AUTO;
        if (s == "cpu") return Device::CPU;
        if (s == "cuda" || s == "hip") return Device::CUDA;
        throw py::value_error("Invalid string value " + s + " for enum Device");
    }));
    py::implicitly_convertible<std::string, Device>();

    // ---- RANSACOptions: Python-side defaults differ from the C++ struct's -------------------
    py::class_<RANSACOptions> PyRANSAC(m, "RANSACOptions");
    PyRANSAC.def(py::init([]() {
        RANSACOptions o;  // /root/reference/pycolmap/optim/bindings.h:10-18
        o.max_error = 4.0;
        o.min_inlier_ratio = 0.01;
        o.confidence = 0.9999;
        o.min_num_trials = 1000;
        o.max_num_trials = 100000;
        return o;
    }));
    PyRANSAC.def_readwrite("max_error", &RANSACOptions::max_error)
        .def_readwrite("min_inlier_ratio", &RANSACOptions::min_inlier_ratio)
        .def_readwrite("confidence", &RANSACOptions::confidence)
        .def_readwrite("dyn_num_trials_multiplier", &RANSACOptions::dyn_num_trials_multiplier)
        .def_readwrite("min_num_trials", &RANSACOptions::min_num_trials)
        .def_readwrite("max_num_trials", &RANSACOptions::max_num_trials);
    MakeDataclass(PyRANSAC, {"max_error", "min_inlier_ratio", "confidence", "dyn_num_trials_multiplier",
                             "min_num_trials", "max_num_trials"});

    py::class_<SiftMatchingOptions> PySift(m, "SiftMatchingOptions");
    PySift.def(py::init<>())
        .def_readwrite("num_threads", &SiftMatchingOptions::num_threads)
        .def_readwrite("gpu_index", &SiftMatchingOptions::gpu_index,
                       "Index of the GPU used for feature matching. For multi-GPU matching, you should "
                       "separate multiple GPU indices by comma, e.g., \"0,1,2,3\".")
        .def_readwrite("max_ratio", &SiftMatchingOptions::max_ratio,
                       "Maximum distance ratio between first and second best match.")
        .def_readwrite("max_distance", &SiftMatchingOptions::max_distance, "Maximum distance to best match.")
        .def_readwrite("cross_check", &SiftMatchingOptions::cross_check,
                       "Whether to enable cross checking in matching.")
        .def_readwrite("max_num_matches", &SiftMatchingOptions::max_num_matches, "Maximum number of matches.")
        .def_readwrite("guided_matching", &SiftMatchingOptions::guided_matching,
                       "Whether to perform guided matching, if geometric verification succeeds.");
    MakeDataclass(PySift, {"num_threads", "gpu_index", "max_ratio", "max_distance", "cross_check",
                           "max_num_matches", "guided_matching"});

    py::class_<ExhaustiveMatchingOptions> PyExh(m, "ExhaustiveMatchingOptions");
    PyExh.def(py::init<>()).def_readwrite("block_size", &ExhaustiveMatchingOptions::block_size);
    MakeDataclass(PyExh, {"block_size"});

    py::class_<SequentialMatchingOptions> PySeq(m, "SequentialMatchingOptions");
    PySeq.def(py::init<>())
        .def_readwrite("overlap", &SequentialMatchingOptions::overlap, "Number of overlapping image pairs.")
        .def_readwrite("quadratic_overlap", &SequentialMatchingOptions::quadratic_overlap,
                       "Whether to match images against their quadratic neighbors.")
        .def_readwrite("loop_detection", &SequentialMatchingOptions::loop_detection)
        .def_readwrite("loop_detection_num_images", &SequentialMatchingOptions::loop_detection_num_images)
        .def_readwrite("loop_detection_num_nearest_neighbors",
                       &SequentialMatchingOptions::loop_detection_num_nearest_neighbors)
        .def_readwrite("loop_detection_num_checks", &SequentialMatchingOptions::loop_detection_num_checks)
        .def_readwrite("loop_detection_num_images_after_verification",
                       &SequentialMatchingOptions::loop_detection_num_images_after_verification)
        .def_readwrite("loop_detection_max_num_features",
                       &SequentialMatchingOptions::loop_detection_max_num_features)
        .def_readwrite("vocab_tree_path", &SequentialMatchingOptions::vocab_tree_path);
    MakeDataclass(PySeq, {"overlap", "quadratic_overlap", "loop_detection", "loop_detection_num_images",
                          "loop_detection_num_nearest_neighbors", "loop_detection_num_checks",
                          "loop_detection_num_images_after_verification", "loop_detection_max_num_features",
                          "vocab_tree_path"});

    // match_vocabtree is outside this library's scope, but its option class exists so that a
    // script written for the reference constructs it without error and fails at the call, with the reason
    // (/root/reference/pycolmap/pipeline/match_features.h:177-214; defaults of COLMAP 3.9.1)
    struct VocabTreeMatchingOptions {
        int num_images = 100, num_nearest_neighbors = 5, num_checks = 256, num_images_after_verification = 0;
        int max_num_features = -1;
        std::string vocab_tree_path, match_list_path;
    };
    py::class_<SpatialMatchingOptions> PySp(m, "SpatialMatchingOptions");
    PySp.def(py::init<>())
        .def_readwrite("is_gps", &SpatialMatchingOptions::is_gps,
                       "Whether the location priors in the database are GPS coordinates in the form of longitude and "
                       "latitude coordinates in degrees.")
        .def_readwrite("ignore_z", &SpatialMatchingOptions::ignore_z,
                       "Whether to ignore the Z-component of the location prior.")
        .def_readwrite("max_num_neighbors", &SpatialMatchingOptions::max_num_neighbors,
                       "The maximum number of nearest neighbors to match.")
        .def_readwrite("max_distance", &SpatialMatchingOptions::max_distance,
                       "The maximum distance between the query and nearest neighbor [meters].");
    MakeDataclass(PySp, {"is_gps", "ignore_z", "max_num_neighbors", "max_distance"});
    py::class_<VocabTreeMatchingOptions> PyVt(m, "VocabTreeMatchingOptions");
    PyVt.def(py::init<>())
        .def_readwrite("num_images", &VocabTreeMatchingOptions::num_images)
        .def_readwrite("num_nearest_neighbors", &VocabTreeMatchingOptions::num_nearest_neighbors)
        .def_readwrite("num_checks", &VocabTreeMatchingOptions::num_checks)
        .def_readwrite("num_images_after_verification", &VocabTreeMatchingOptions::num_images_after_verification)
        .def_readwrite("max_num_features", &VocabTreeMatchingOptions::max_num_features)
        .def_readwrite("vocab_tree_path", &VocabTreeMatchingOptions::vocab_tree_path)
        .def_readwrite("match_list_path", &VocabTreeMatchingOptions::match_list_path);
    MakeDataclass(PyVt, {"num_images", "num_nearest_neighbors", "num_checks", "num_images_after_verification",
                         "max_num_features", "vocab_tree_path", "match_list_path"});

    py::class_<TwoViewGeometryOptions> PyTvgO(m, "TwoViewGeometryOptions");
    PyTvgO.def(py::init<>())  // C++ defaults, incl. the C++ RANSAC defaults
        .def_readwrite("min_num_inliers", &TwoViewGeometryOptions::min_num_inliers)
        .def_readwrite("min_E_F_inlier_ratio", &TwoViewGeometryOptions::min_E_F_inlier_ratio)
        .def_readwrite("max_H_inlier_ratio", &TwoViewGeometryOptions::max_H_inlier_ratio)
        .def_readwrite("watermark_min_inlier_ratio", &TwoViewGeometryOptions::watermark_min_inlier_ratio)
        .def_readwrite("watermark_border_size", &TwoViewGeometryOptions::watermark_border_size)
        .def_readwrite("detect_watermark", &TwoViewGeometryOptions::detect_watermark)
        .def_readwrite("multiple_ignore_watermark", &TwoViewGeometryOptions::multiple_ignore_watermark)
        .def_readwrite("force_H_use", &TwoViewGeometryOptions::force_H_use)
        .def_readwrite("compute_relative_pose", &TwoViewGeometryOptions::compute_relative_pose)
        .def_readwrite("multiple_models", &TwoViewGeometryOptions::multiple_models)
        .def_readwrite("ransac", &TwoViewGeometryOptions::ransac_options);
    MakeDataclass(PyTvgO, {"min_num_inliers", "min_E_F_inlier_ratio", "max_H_inlier_ratio",
                           "watermark_min_inlier_ratio", "watermark_border_size", "detect_watermark",
                           "multiple_ignore_watermark", "force_H_use", "compute_relative_pose",
                           "multiple_models", "ransac"});

    // Rotation3d / Rigid3d: value types of cam2_from_cam1 (/root/reference/pycolmap/geometry/bindings.h:24-104)
    py::class_<PyRotation3d>(m, "Rotation3d")
        .def(py::init<>())
        .def(py::init([](const std::array<double, 4>& xyzw) {
                 PyRotation3d r;
                 r.xyzw = xyzw;
                 return r;
             }),
             "xyzw"_a, "Quaternion in [x,y,z,w] format.")
        .def(py::init([](const py::array_t<double, py::array::c_style | py::array::forcecast>& a) {
                 // a 3 x 3 rotation matrix or an axis-angle 3-vector (the reference's two other constructors)
                 if (a.ndim() == 2 && a.shape(0) == 3 && a.shape(1) == 3) {
                     std::array<double, 9> m;
                     std::memcpy(m.data(), a.data(), sizeof(double) * 9);
                     return PyRotation3d::FromMatrix(m);
                 }
                 if (a.ndim() == 1 && a.shape(0) == 3) return PyRotation3d::FromAxisAngle({{a.at(0), a.at(1), a.at(2)}});
                 if (a.ndim() == 1 && a.shape(0) == 4) {
                     PyRotation3d r;
                     r.xyzw = {{a.at(0), a.at(1), a.at(2), a.at(3)}};
                     return r;
                 }
                 throw py::value_error("Rotation3d: expected a quaternion [x,y,z,w], a 3 x 3 rotation matrix or an axis-angle 3-vector");
             }),
             "rotmat_or_axis_angle"_a, "3x3 rotation matrix, or axis-angle 3D vector.")
        .def("__mul__", [](const PyRotation3d& a, const PyRotation3d& b) { return a.Mul(b); }, py::is_operator())
        .def("__mul__",
             [](const PyRotation3d& r, const py::array_t<double, py::array::c_style | py::array::forcecast>& v) -> py::array_t<double> {
                 if (v.ndim() == 1 && v.shape(0) == 3) {
                     const std::array<double, 3> o = r.Rotate({{v.at(0), v.at(1), v.at(2)}});
                     py::array_t<double> out(3);
                     std::memcpy(out.mutable_data(), o.data(), sizeof(double) * 3);
                     return out;
                 }
                 if (v.ndim() == 2 && v.shape(1) == 3) {  // points * R^T
                     const std::array<double, 9> R = r.Matrix();
                     py::array_t<double> out({v.shape(0), static_cast<py::ssize_t>(3)});
                     for (py::ssize_t i = 0; i < v.shape(0); ++i)
                         for (int j = 0; j < 3; ++j)
                             out.mutable_at(i, j) = v.at(i, 0) * R[3 * j] + v.at(i, 1) * R[3 * j + 1] + v.at(i, 2) * R[3 * j + 2];
                     return out;
                 }
                 throw py::value_error("Rotation3d * x: x must be a Rotation3d, a 3-vector or an N x 3 array");
             },
             py::is_operator())
        .def("normalize",
             [](PyRotation3d& r) {
                 const double n = std::sqrt(r.SquaredNorm());
                 for (double& c : r.xyzw) c /= n;
             })
        .def("angle", &PyRotation3d::Angle)
        .def("angle_to", &PyRotation3d::AngleTo, "other"_a)
        .def("inverse", &PyRotation3d::Inverse)
        .def_property(
            "quat",
            [](const PyRotation3d& r) {
                py::array_t<double> a(4);
                std::memcpy(a.mutable_data(), r.xyzw.data(), sizeof(double) * 4);
                return a;
            },
            [](PyRotation3d& r, const std::array<double, 4>& q) { r.xyzw = q; }, "Quaternion in [x,y,z,w] format.")
        .def("matrix", [](const PyRotation3d& r) { return Mat3(r.Matrix()); })
        .def("norm",
             [](const PyRotation3d& r) {
                 return std::sqrt(r.xyzw[0] * r.xyzw[0] + r.xyzw[1] * r.xyzw[1] + r.xyzw[2] * r.xyzw[2] +
                                  r.xyzw[3] * r.xyzw[3]);
             })
        .def("__repr__", [](const PyRotation3d& r) {
            std::ostringstream ss;
            ss << "Rotation3d(quat_xyzw=[" << r.xyzw[0] << ", " << r.xyzw[1] << ", " << r.xyzw[2] << ", " << r.xyzw[3]
               << "])";
            return ss.str();
        });
    py::class_<PyRigid3d>(m, "Rigid3d")
        .def(py::init<>())
        .def(py::init([](const PyRotation3d& r, const std::array<double, 3>& t) {
            PyRigid3d x;
            x.rotation = r;
            x.translation = t;
            return x;
        }))
        .def(py::init([](const py::array_t<double, py::array::c_style | py::array::forcecast>& a) {
                 if (a.ndim() != 2 || a.shape(0) != 3 || a.shape(1) != 4) throw py::value_error("Rigid3d: expected a 3 x 4 matrix [R | t]");
                 std::array<double, 9> m;
                 PyRigid3d x;
                 for (int i = 0; i < 3; ++i) {
                     for (int j = 0; j < 3; ++j) m[3 * i + j] = a.at(i, j);
                     x.translation[i] = a.at(i, 3);
                 }
                 x.rotation = PyRotation3d::FromMatrix(m);
                 return x;
             }),
             "matrix"_a)
        .def("__mul__", [](const PyRigid3d& a, const PyRigid3d& b) { return a.Mul(b); }, py::is_operator())
        .def("__mul__",
             [](const PyRigid3d& r, const py::array_t<double, py::array::c_style | py::array::forcecast>& v) -> py::array_t<double> {
                 if (v.ndim() == 1 && v.shape(0) == 3) {
                     const std::array<double, 3> o = r.Apply({{v.at(0), v.at(1), v.at(2)}});
                     py::array_t<double> out(3);
                     std::memcpy(out.mutable_data(), o.data(), sizeof(double) * 3);
                     return out;
                 }
                 if (v.ndim() == 2 && v.shape(1) == 3) {  // points * R^T, + t to every row
                     const std::array<double, 9> R = r.rotation.Matrix();
                     py::array_t<double> out({v.shape(0), static_cast<py::ssize_t>(3)});
                     for (py::ssize_t i = 0; i < v.shape(0); ++i)
                         for (int j = 0; j < 3; ++j)
                             out.mutable_at(i, j) =
                                 (v.at(i, 0) * R[3 * j] + v.at(i, 1) * R[3 * j + 1] + v.at(i, 2) * R[3 * j + 2]) + r.translation[j];
                     return out;
                 }
                 throw py::value_error("Rigid3d * x: x must be a Rigid3d, a 3-vector or an N x 3 array");
             },
             py::is_operator())
        .def("inverse", &PyRigid3d::Inverse)
        .def("essential_matrix",
             [](const PyRigid3d& r) {  // EssentialMatrixFromPose: [t / |t|]_x R
                 const double n = std::sqrt(r.translation[0] * r.translation[0] + r.translation[1] * r.translation[1] +
                                            r.translation[2] * r.translation[2]);
                 std::array<double, 3> t = r.translation;
                 if (n > 0.0)
                     for (double& c : t) c /= n;
                 const std::array<double, 9> R = r.rotation.Matrix();
                 const double X[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
                 std::array<double, 9> E{};
                 for (int i = 0; i < 3; ++i)
                     for (int j = 0; j < 3; ++j) E[3 * i + j] = X[3 * i] * R[j] + X[3 * i + 1] * R[3 + j] + X[3 * i + 2] * R[6 + j];
                 return Mat3(E);
             })
        .def_static(
            "interpolate",
            [](const PyRigid3d& a, const PyRigid3d& b, double t) {  // InterpolateCameraPoses: slerp + linear translation
                PyRigid3d r;
                r.rotation = a.rotation.Slerp(t, b.rotation);
                for (int i = 0; i < 3; ++i) r.translation[i] = a.translation[i] + (b.translation[i] - a.translation[i]) * t;
                return r;
            },
            "cam_from_world1"_a, "cam_from_world2"_a, "t"_a)
        .def_readwrite("rotation", &PyRigid3d::rotation)
        .def_property(
            "translation",
            [](const PyRigid3d& r) {
                py::array_t<double> a(3);
                std::memcpy(a.mutable_data(), r.translation.data(), sizeof(double) * 3);
                return a;
            },
            [](PyRigid3d& r, const std::array<double, 3>& t) { r.translation = t; })
        .def("matrix",
             [](const PyRigid3d& r) {  // Rigid3d::ToMatrix: [R | t], 3 x 4
                 const std::array<double, 9> R = r.rotation.Matrix();
                 py::array_t<double> a({3, 4});
                 double* d = a.mutable_data();
                 for (int i = 0; i < 3; ++i) {
                     for (int j = 0; j < 3; ++j) d[4 * i + j] = R[3 * i + j];
                     d[4 * i + 3] = r.translation[i];
                 }
                 return a;
             })
        .def("__repr__", [](const PyRigid3d& r) {
            std::ostringstream ss;
            ss << "Rigid3d(quat_xyzw=[" << r.rotation.xyzw[0] << ", " << r.rotation.xyzw[1] << ", " << r.rotation.xyzw[2]
               << ", " << r.rotation.xyzw[3] << "], t=[" << r.translation[0] << ", " << r.translation[1] << ", "
               << r.translation[2] << "])";
            return ss.str();
        });

    py::class_<PyTwoViewGeometry> PyTvg(m, "TwoViewGeometry");
    py::object cfg_enum = py::module_::import("enum").attr("IntEnum")(
        "TwoViewGeometryConfiguration",
        py::dict("UNDEFINED"_a = 0, "DEGENERATE"_a = 1, "CALIBRATED"_a = 2, "UNCALIBRATED"_a = 3, "PLANAR"_a = 4,
                 "PANORAMIC"_a = 5, "PLANAR_OR_PANORAMIC"_a = 6, "WATERMARK"_a = 7, "MULTIPLE"_a = 8));
    m.attr("TwoViewGeometryConfiguration") = cfg_enum;
    PyTvg.def(py::init<>())
        .def_property_readonly("config", [cfg_enum](const PyTwoViewGeometry& s) { return cfg_enum(s.config); })
        .def_property_readonly("E", [](const PyTwoViewGeometry& s) { return Mat3(s.E); })
        .def_property_readonly("F", [](const PyTwoViewGeometry& s) { return Mat3(s.F); })
        .def_property_readonly("H", [](const PyTwoViewGeometry& s) { return Mat3(s.H); })
        .def_readonly("cam2_from_cam1", &PyTwoViewGeometry::cam2_from_cam1)
        .def_property_readonly("inlier_matches",
                               [](const PyTwoViewGeometry& s) { return MatchesArray(s.inlier_matches); })
        .def_readonly("tri_angle", &PyTwoViewGeometry::tri_angle)
        .def("invert",
             [](PyTwoViewGeometry& g) {  // TwoViewGeometry::Invert: the geometry as seen from image 2
                 TwoViewGeometryRow r;
                 r.config = g.config;
                 r.E = g.E;
                 r.F = g.F;
                 r.H = g.H;
                 r.inlier_matches = std::move(g.inlier_matches);
                 r.qvec = {{g.cam2_from_cam1.rotation.xyzw[3], g.cam2_from_cam1.rotation.xyzw[0],
                            g.cam2_from_cam1.rotation.xyzw[1], g.cam2_from_cam1.rotation.xyzw[2]}};
                 r.tvec = g.cam2_from_cam1.translation;
                 r.Invert();
                 g.E = r.E;
                 g.F = r.F;
                 g.H = r.H;
                 g.inlier_matches = std::move(r.inlier_matches);
                 g.cam2_from_cam1.rotation.xyzw = {{r.qvec[1], r.qvec[2], r.qvec[3], r.qvec[0]}};
                 g.cam2_from_cam1.translation = r.tvec;
             })
        .def("todict",
             [cfg_enum](const PyTwoViewGeometry& s) {
                 return py::dict("config"_a = cfg_enum(s.config), "E"_a = Mat3(s.E), "F"_a = Mat3(s.F), "H"_a = Mat3(s.H),
                                 "cam2_from_cam1"_a = s.cam2_from_cam1, "inlier_matches"_a = MatchesArray(s.inlier_matches),
                                 "tri_angle"_a = s.tri_angle);
             })
        .def("__copy__", [](const PyTwoViewGeometry& s) { return PyTwoViewGeometry(s); })
        .def("__deepcopy__", [](const PyTwoViewGeometry& s, const py::dict&) { return PyTwoViewGeometry(s); });

    // ---- Camera (estimators.h), Image ---------------------------------------------------------
    BindCamera(m);
    // Image: the database-facing part of /root/reference/pycolmap/scene/image.h:74-130 (identifiers, name, pose and pose
    // prior, the keypoints it was constructed with); the reconstruction bookkeeping (points3D, observations) belongs
    // to COLMAP's mapper and is not part of this library
    struct PyImage {
        uint32_t image_id = 0xFFFFFFFFu, camera_id = 0xFFFFFFFFu;  // kInvalidImageId / kInvalidCameraId
        std::string name;
        PyRigid3d cam_from_world, cam_from_world_prior;
        std::vector<std::array<double, 2>> keypoints;
        PyImage() {
            const double nan = std::nan("");  // Image(): the prior is "unknown"
            cam_from_world_prior.rotation.xyzw = {{nan, nan, nan, nan}};
            cam_from_world_prior.translation = {{nan, nan, nan}};
        }
    };
    auto image_from_row = [](const ImageRow& r) {
        PyImage im;
        im.image_id = r.image_id;
        im.camera_id = r.camera_id;
        im.name = r.name;
        im.cam_from_world_prior.rotation.xyzw = {{r.prior_q[1], r.prior_q[2], r.prior_q[3], r.prior_q[0]}};
        im.cam_from_world_prior.translation = r.prior_t;
        return im;
    };
    auto row_from_image = [](const PyImage& im) {
        ImageRow r;
        r.image_id = im.image_id;
        r.camera_id = im.camera_id;
        r.name = im.name;
        const auto& q = im.cam_from_world_prior.rotation.xyzw;
        r.prior_q = {{q[3], q[0], q[1], q[2]}};
        r.prior_t = im.cam_from_world_prior.translation;
        return r;
    };
    auto camera_from_row = [](const CameraRow& r) {
        PyCamera c;
        c.camera_id = r.camera_id;
        c.model = r.model_id;
        c.width = r.width;
        c.height = r.height;
        c.params = r.params;
        c.has_prior_focal_length = r.has_prior_focal_length;
        return c;
    };
    auto row_from_camera = [](const PyCamera& c) {
        c.CheckParams();
        CameraRow r;
        r.camera_id = c.camera_id;
        r.model_id = c.model;
        r.width = c.width;
        r.height = c.height;
        r.params = c.params;
        r.has_prior_focal_length = c.has_prior_focal_length;
        return r;
    };
    py::class_<PyImage>(m, "Image")
        .def(py::init<>())
        .def(py::init([](const std::string& name, const std::vector<std::array<double, 2>>& keypoints,
                         const PyRigid3d& cam_from_world, uint32_t camera_id, uint32_t id) {
                 PyImage im;
                 im.name = name;
                 im.keypoints = keypoints;
                 im.cam_from_world = cam_from_world;
                 im.camera_id = camera_id;
                 im.image_id = id;
                 return im;
             }),
             "name"_a = "", "keypoints"_a = std::vector<std::array<double, 2>>(), "cam_from_world"_a = PyRigid3d(),
             "camera_id"_a = 0xFFFFFFFFu, "id"_a = 0xFFFFFFFFu)
        .def_readwrite("image_id", &PyImage::image_id, "Unique identifier of image.")
        .def_property(
            "camera_id", [](const PyImage& im) { return im.camera_id; },
            [](PyImage& im, uint32_t id) {
                if (id == 0xFFFFFFFFu) throw py::value_error(CheckMessage(__FILE__, __LINE__, "camera_id != kInvalidCameraId"));
                im.camera_id = id;
            },
            "Unique identifier of the camera.")
        .def_readwrite("name", &PyImage::name, "Name of the image.")
        .def_readwrite("cam_from_world", &PyImage::cam_from_world,
                       "The pose of the image, defined as the transformation from world to camera space.")
        .def_readwrite("cam_from_world_prior", &PyImage::cam_from_world_prior,
                       "The pose prior of the image, e.g. extracted from EXIF tags.")
        .def("has_camera", [](const PyImage& im) { return im.camera_id != 0xFFFFFFFFu; },
             "Check whether identifier of camera has been set.")
        .def("num_points2D", [](const PyImage& im) { return im.keypoints.size(); },
             "Get the number of image points (keypoints).")
        .def("__copy__", [](const PyImage& im) { return PyImage(im); })
        .def("__deepcopy__", [](const PyImage& im, const py::dict&) { return PyImage(im); })
        .def("__repr__", [](const PyImage& im) {
            std::ostringstream ss;
            ss << "Image(image_id=" << (im.image_id != 0xFFFFFFFFu ? std::to_string(im.image_id) : "Invalid")
               << ", camera_id=" << (im.camera_id != 0xFFFFFFFFu ? std::to_string(im.camera_id) : "Invalid") << ", name=\""
               << im.name << "\", triangulated=0/" << im.keypoints.size() << ")";
            return ss.str();
        });

    // ---- Database ---------------------------------------------------------------------------
    py::class_<Database>(m, "Database")
        .def(py::init([](const py::object& path) {
                 // Database::Open: creates the file and COLMAP's tables when they are missing
                 return std::make_unique<Database>(PathToString(path));
             }),
             "path"_a)
        .def("open", [](Database& db, const py::object& path) { db.Open(PathToString(path)); }, "path"_a)
        .def("close", &Database::Close)
        .def("num_keypoints_for_image", &Database::NumKeypointsForImage, "image_id"_a)
        .def("num_descriptors_for_image", &Database::NumDescriptorsForImage, "image_id"_a)
        .def("exists_camera", &Database::ExistsCamera, "camera_id"_a)
        .def("exists_image", &Database::ExistsImage, "image_id"_a)
        .def("read_camera", [camera_from_row](const Database& db, camera_t id) { return camera_from_row(db.ReadCamera(id)); },
             "camera_id"_a)
        .def("read_all_cameras",
             [camera_from_row](const Database& db) {
                 std::vector<PyCamera> out;
                 for (const CameraRow& r : db.ReadAllCameras()) out.push_back(camera_from_row(r));
                 return out;
             })
        .def("read_image", [image_from_row](const Database& db, image_t id) { return image_from_row(db.ReadImage(id)); },
             "image_id"_a)
        .def("read_image_with_name",
             [image_from_row](const Database& db, const std::string& name) { return image_from_row(db.ReadImageWithName(name)); },
             "name"_a)
        .def("read_all_images",
             [image_from_row](const Database& db) {
                 std::vector<PyImage> out;
                 for (const ImageRow& r : db.ReadAllImages()) out.push_back(image_from_row(r));
                 return out;
             })
        .def("write_camera",
             [row_from_camera](Database& db, const PyCamera& c, bool use_camera_id) {
                 return db.WriteCamera(row_from_camera(c), use_camera_id);
             },
             "camera"_a, "use_camera_id"_a = false, "Returns the camera_id of the new row.")
        .def("write_image",
             [row_from_image](Database& db, const PyImage& im, bool use_image_id) {
                 return db.WriteImage(row_from_image(im), use_image_id);
             },
             "image"_a, "use_image_id"_a = false, "Returns the image_id of the new row.")
        .def_property_readonly("num_cameras", &Database::NumCameras)
        .def_property_readonly("num_images", &Database::NumImages)
        .def_property_readonly("num_keypoints", &Database::NumKeypoints)
        .def_property_readonly("num_descriptors", &Database::NumDescriptors)
        .def_property_readonly("num_matches", &Database::NumMatches)
        .def_property_readonly("num_inlier_matches", &Database::NumInlierMatches)
        .def_property_readonly("num_matched_image_pairs", &Database::NumMatchedImagePairs)
        .def_property_readonly("num_verified_image_pairs", &Database::NumVerifiedImagePairs)
        .def_static("image_pair_to_pair_id", &Database::ImagePairToPairId, "image_id1"_a, "image_id2"_a)
        .def_static("pair_id_to_image_pair",
                    [](image_pair_t pid) {
                        image_t a, b;
                        Database::PairIdToImagePair(pid, &a, &b);
                        return std::make_pair(a, b);
                    },
                    "pair_id"_a)
        .def("exists_matches", &Database::ExistsMatches, "image_id1"_a, "image_id2"_a)
        .def("exists_inlier_matches", &Database::ExistsInlierMatches, "image_id1"_a, "image_id2"_a)
        .def("read_matches",
             [](const Database& db, image_t a, image_t b) { return MatchesArray(db.ReadMatches(a, b)); },
             "image_id1"_a, "image_id2"_a)
        // keypoints / descriptors / matches accessors: the reference leaves them unbound
        // (/root/reference/pycolmap/scene/database.h:35-41); names follow COLMAP's Database methods
        .def("set_bulk_write_mode", &Database::SetBulkWriteMode, "on"_a,
             "Rollback journal instead of WAL while appending many blobs; returns the journal mode in effect. "
             "WAL (COLMAP's mode) is restored when switched off or on close.")
        .def("exists_keypoints", &Database::ExistsKeypoints, "image_id"_a)
        .def("exists_descriptors", &Database::ExistsDescriptors, "image_id"_a)
        .def("read_keypoints",
             [](const Database& db, image_t id) {
                 uint32_t rows = 0, cols = 0;
                 const std::vector<float> v = db.ReadKeypoints(id, &rows, &cols);
                 py::array_t<float> a({static_cast<py::ssize_t>(rows), static_cast<py::ssize_t>(cols)});
                 if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(float));
                 return a;
             },
             "image_id"_a, "rows x cols float32 (cols = 2, 4 or 6: x, y[, scale, orientation | a11, a12, a21, a22])")
        .def("read_descriptors",
             [](const Database& db, image_t id) {
                 uint32_t rows = 0;
                 const std::vector<uint8_t> v = db.ReadDescriptors(id, &rows);
                 py::array_t<uint8_t> a({static_cast<py::ssize_t>(rows), static_cast<py::ssize_t>(128)});
                 if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size());
                 return a;
             },
             "image_id"_a, "rows x 128 uint8")
        .def("write_keypoints",
             [](Database& db, image_t id, const py::array_t<float, py::array::c_style | py::array::forcecast>& kp) {
                 if (kp.ndim() != 2 || (kp.shape(1) != 2 && kp.shape(1) != 4 && kp.shape(1) != 6))
                     throw py::value_error("keypoints must be an N x 2, N x 4 or N x 6 float32 array");
                 db.WriteKeypoints(id, kp.data(), static_cast<uint32_t>(kp.shape(0)), static_cast<uint32_t>(kp.shape(1)));
             },
             "image_id"_a, "keypoints"_a)
        .def("write_descriptors",
             [](Database& db, image_t id, const py::array_t<uint8_t, py::array::c_style>& d) {
                 if (d.ndim() != 2 || d.shape(1) != 128)
                     throw py::value_error("descriptors must be an N x 128 uint8 array");
                 db.WriteDescriptors(id, d.data(), static_cast<uint32_t>(d.shape(0)));
             },
             "image_id"_a, "descriptors"_a)
        .def("write_matches",
             [](Database& db, image_t a, image_t b,
                const py::array_t<uint32_t, py::array::c_style | py::array::forcecast>& m) {
                 if (m.size() != 0 && (m.ndim() != 2 || m.shape(1) != 2))
                     throw py::value_error("matches must be an M x 2 unsigned integer array");
                 db.WriteMatches(a, b, std::vector<uint32_t>(m.data(), m.data() + m.size()));
             },
             "image_id1"_a, "image_id2"_a, "matches"_a)
        .def("delete_matches", &Database::DeleteMatches, "image_id1"_a, "image_id2"_a)
        .def("delete_inlier_matches", &Database::DeleteInlierMatches, "image_id1"_a, "image_id2"_a)
        .def("read_two_view_geometry",
             [](const Database& db, image_t a, image_t b) {
                 const TwoViewGeometryRow r = db.ReadTwoViewGeometry(a, b);
                 PyTwoViewGeometry g;
                 g.config = r.config;
                 g.E = r.E;
                 g.F = r.F;
                 g.H = r.H;
                 g.inlier_matches = r.inlier_matches;
                 g.cam2_from_cam1.rotation.xyzw = {{r.qvec[1], r.qvec[2], r.qvec[3], r.qvec[0]}};
                 g.cam2_from_cam1.translation = r.tvec;
                 return g;
             },
             "image_id1"_a, "image_id2"_a);

    // DatabaseTransaction (/root/reference/pycolmap/scene/database.h:44-45): BEGIN on construction, END when the
    // object goes away - COLMAP's scope guard as Python sees it; also usable as a context manager, where an exception
    // rolls back (an extension)
    struct PyDbTransaction {
        Database* db;
        bool open = true;
        explicit PyDbTransaction(Database* d) : db(d) { db->BeginTransaction(); }
        PyDbTransaction(const PyDbTransaction&) = delete;
        void End(bool commit) {
            if (!open) return;
            open = false;
            if (commit) db->EndTransaction(); else db->RollbackTransaction();
        }
        ~PyDbTransaction() {
            try {
                End(true);
            } catch (...) {
            }
        }
    };
    py::class_<PyDbTransaction>(m, "DatabaseTransaction")
        .def(py::init<Database*>(), "database"_a, py::keep_alive<1, 2>())
        .def("__enter__", [](PyDbTransaction& t) -> PyDbTransaction& { return t; }, py::return_value_policy::reference)
        .def("__exit__", [](PyDbTransaction& t, const py::object& type, const py::object&, const py::object&) {
            t.End(type.is_none());
            return false;
        });

    // ---- single-pair estimators (estimators.h) ----------------------------------------------------
    BindEstimators(m);

    // ---- pipeline entry points ----------------------------------------------------------------
    m.def("_parse_gpu_index", &ParseGpuIndex, "gpu_index"_a, "SiftMatchingOptions.gpu_index -> device list (test hook)");
    auto run_pipeline = [](const py::object& database_path, const SiftMatchingOptions& sift,
                           const TwoViewGeometryOptions& tvg, Device device,
                           const std::function<void(MatchController&)>& body) {
        const std::string db_path = PathToString(database_path);
        AMC_THROW_CHECK_FILE_EXISTS(db_path);
        RequireAccelerator(device);
        const auto t0 = std::chrono::steady_clock::now();
        auto since = [](std::chrono::steady_clock::time_point t) {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
        };
        auto ctrl = std::make_unique<MatchController>(db_path, sift, tvg, ParseGpuIndex(sift.gpu_index));
        RunInterruptible(*ctrl, [&] {
            ctrl->Setup();
            body(*ctrl);
        });
        py::dict st = StatsDict(ctrl->stats);
        const auto t1 = std::chrono::steady_clock::now();
        {
            py::gil_scoped_release release;
            ctrl.reset();  // closes the database (WAL again), frees the arena and the contexts' buffers
        }
        st["teardown_ms"] = since(t1);
        st["call_ms"] = since(t0);  // the whole call as the caller's clock sees it
        py::module_::import("pycolmap_amd._pycolmap").attr("_last_stats") = st;
    };

    m.def(
        "match_exhaustive",
        [run_pipeline](const py::object& database_path, const SiftMatchingOptions& sift,
                       const ExhaustiveMatchingOptions& mo, const TwoViewGeometryOptions& tvg, Device device) {
            run_pipeline(database_path, sift, tvg, device, [&](MatchController& c) { RunExhaustive(c, mo); });
        },
        "database_path"_a, "sift_options"_a = SiftMatchingOptions(),
        "matching_options"_a = ExhaustiveMatchingOptions(), "verification_options"_a = TwoViewGeometryOptions(),
        "device"_a = Device::AUTO, "Exhaustive feature matching");
    m.def(
        "match_sequential",
        [run_pipeline](const py::object& database_path, const SiftMatchingOptions& sift,
                       const SequentialMatchingOptions& mo, const TwoViewGeometryOptions& tvg, Device device) {
            // options of COLMAP's vocabulary-tree retrieval that have no counterpart in the feature-voting retrieval
            // used here (controller.cc): accepted for drop-in compatibility, and said out loud when they are set
            const SequentialMatchingOptions def;
            if (mo.loop_detection && (!mo.vocab_tree_path.empty() ||
                                      mo.loop_detection_num_nearest_neighbors != def.loop_detection_num_nearest_neighbors ||
                                      mo.loop_detection_num_checks != def.loop_detection_num_checks ||
                                      mo.loop_detection_num_images_after_verification != def.loop_detection_num_images_after_verification))
                Logging::Write(Logging::WARNING, "match_sequential", 0,
                               "vocab_tree_path / loop_detection_num_nearest_neighbors / _num_checks / "
                               "_num_images_after_verification are ignored: loop closure candidates come from exact feature "
                               "voting on the first loop_detection_max_num_features descriptors, not from a vocabulary tree");
            run_pipeline(database_path, sift, tvg, device, [&](MatchController& c) { RunSequential(c, mo); });
        },
        "database_path"_a, "sift_options"_a = SiftMatchingOptions(),
        "matching_options"_a = SequentialMatchingOptions(), "verification_options"_a = TwoViewGeometryOptions(),
        "device"_a = Device::AUTO, "Sequential feature matching");
    m.def(
        "match_spatial",
        [run_pipeline](const py::object& database_path, const SiftMatchingOptions& sift,
                       const SpatialMatchingOptions& mo, const TwoViewGeometryOptions& tvg, Device device) {
            run_pipeline(database_path, sift, tvg, device, [&](MatchController& c) { RunSpatial(c, mo); });
        },
        "database_path"_a, "sift_options"_a = SiftMatchingOptions(),
        "matching_options"_a = SpatialMatchingOptions(), "verification_options"_a = TwoViewGeometryOptions(),
        "device"_a = Device::AUTO, "Spatial feature matching");
    m.def(
        "verify_matches",
        [run_pipeline](const py::object& database_path, const py::object& pairs_path,
                       const TwoViewGeometryOptions& tvg) {
            const std::string pp = PathToString(pairs_path);
            const std::string dbp = PathToString(database_path);
            AMC_THROW_CHECK_FILE_EXISTS(dbp);
            AMC_THROW_CHECK_FILE_EXISTS(pp);
            run_pipeline(database_path, SiftMatchingOptions(), tvg, Device::AUTO,
                         [&](MatchController& c) { RunImagePairs(c, pp); });
        },
        "database_path"_a, "pairs_path"_a, "options"_a = TwoViewGeometryOptions(),
        "Run geometric verification of the matches");
    auto unsupported = [](const char* what) {
        return [what](const py::args&, const py::kwargs&) {
            throw py::value_error(std::string(what) +
                                  " needs a FLANN vocabulary-tree file and is outside pycolmap_amd's scope "
                                  "(SURVEY.md section 8f); use match_exhaustive, match_sequential, match_spatial or verify_matches.");
        };
    };
    m.def("_exhaustive_blocks", &ExhaustiveBlocks, "image_ids"_a, "block_size"_a,
          "Pair blocks of ExhaustiveFeatureMatcher::Run (test hook)");
    m.def("_sequential_blocks", &SequentialBlocks, "ordered_image_ids"_a, "overlap"_a, "quadratic_overlap"_a,
          "Pair blocks of SequentialFeatureMatcher::Run (test hook)");
    m.def("_spatial_blocks", &SpatialBlocks, "image_ids"_a, "priors"_a, "options"_a,
          "Pair blocks of SpatialFeatureMatcher::Run (test hook)");
    m.def("_ell_to_xyz", &EllToXYZ, "lat_lon_alt"_a, "GPSTransform(WGS84)::EllToXYZ of one point (test hook)");
    m.def("match_vocabtree", unsupported("match_vocabtree"));
    m.attr("_last_stats") = py::dict();
    m.def("last_run_stats", []() { return py::module_::import("pycolmap_amd._pycolmap").attr("_last_stats"); },
          "Timing / counters of the most recent match_* / verify_matches call (pycolmap_amd extension).");
}
