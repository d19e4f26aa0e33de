// adapter/ecto_cells.hpp -- the detection cells of wg-perception/tod's plasm, re-hosted on libtodhip.
//
// Same cell names, parameter names, tendril names and tendril types as the reference cells
//   tod::DescriptorMatcher   src/detection/DescriptorMatcher.cpp:58-269
//   tod::GuessGenerator      src/detection/GuessGenerator.cpp:69-276
// and as the third-party cell the reference's graph instantiates for stage A
//   ecto_opencv.features2d.FeatureDescriptor   python/object_recognition_tod/detector.py:10,27,35-36,41,44,71-74,80-81
// so that python/object_recognition_tod/detector.py and conf/*.ork run unchanged.
// The cells hold no algorithm: they convert tendril types to the flat buffers of include/todhip.h.
//
// The including translation unit provides the framework types first:
//   * a ROS/ORK build includes <ecto/ecto.hpp>, <opencv2/core/core.hpp>, <opencv2/features2d/features2d.hpp>,
//     ORK-core's ModelReader.h / pose_result.h and defines TOD_AMD_WITH_ORK;
//   * this repo's test includes tests/mini_ecto/mini_ecto.hpp (a test double, not part of the product).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/todhip.h"

namespace tod_amd {

// ---- the two JSON lookups the cells need ("radius": 35, "type": "LSH"); no dependency on json_spirit
inline bool json_find(const std::string& js, const std::string& key, size_t* value_pos) {
  const std::string pat = "\"" + key + "\"";
  size_t p = js.find(pat);
  if (p == std::string::npos) return false;
  p = js.find(':', p + pat.size());
  if (p == std::string::npos) return false;
  ++p;
  while (p < js.size() && (js[p] == ' ' || js[p] == '\t' || js[p] == '\n')) ++p;
  *value_pos = p;
  return true;
}
inline double json_number(const std::string& js, const std::string& key, double dflt) {
  size_t p;
  if (!json_find(js, key, &p)) return dflt;
  return std::strtod(js.c_str() + p, nullptr);
}
inline std::string json_string(const std::string& js, const std::string& key, const std::string& dflt) {
  size_t p;
  if (!json_find(js, key, &p) || p >= js.size() || js[p] != '"') return dflt;
  const size_t e = js.find('"', p + 1);
  return e == std::string::npos ? dflt : js.substr(p + 1, e - p - 1);
}

typedef std::string ObjectId;   // object_recognition_core::db::ObjectId

// One shared context per process (one HIP device + stream) for the cells that keep no state in it (FeatureDescriptor,
// GuessGenerator: host-buffer calls only); cells of one plasm run sequentially. A DescriptorMatcher owns a context of its own:
// the object DB and its search settings (ratio test, LSH mode) live there, and two matcher cells with different
// search_json_params must not overwrite each other's.
inline int context_device() { const char* dev = std::getenv("TODHIP_DEVICE"); return dev ? std::atoi(dev) : 0; }
inline todhip_ctx* shared_context() {
  static todhip_ctx* ctx = nullptr;
  if (!ctx) {
    if (todhip_create(context_device(), nullptr, &ctx) != TODHIP_OK)
      throw std::runtime_error("todhip_create failed: no usable MI355X / HIP device");
  }
  return ctx;
}

// ------------------------------------------------------------------------------------------------------------
struct DescriptorMatcher
#ifdef TOD_AMD_WITH_ORK
    : public object_recognition_core::db::bases::ModelReaderBase
#endif
{
  // DescriptorMatcher.cpp:131-140
  static void declare_params(ecto::tendrils& p) {
#ifdef TOD_AMD_WITH_ORK
    object_recognition_core::db::bases::declare_params_impl(p, "TOD");
#endif
    p.declare<std::string>("search_json_params",
                           "JSON string that can contain the following fields: \"radius\" (for epsilon nearest "
                           "neighbor search), \"ratio\" when applying the ratio criterion like in SIFT").required(true);
  }
  // DescriptorMatcher.cpp:142-152
  static void declare_io(const ecto::tendrils&, ecto::tendrils& inputs, ecto::tendrils& outputs) {
    inputs.declare<cv::Mat>("descriptors", "The descriptors to match to the database");
    outputs.declare<std::vector<std::vector<cv::DMatch> > >("matches", "The matches for the input descriptors");
    outputs.declare<std::vector<cv::Mat> >("matches_3d", "For each point, the 3d position of the matches, 1 by n "
                                                         "matrix with 3 channels for, x, y, and z.");
    outputs.declare<std::vector<ObjectId> >("object_ids", "The ids of the objects");
    outputs.declare<std::map<ObjectId, float> >("spans", "The ids of the objects");
  }
  // DescriptorMatcher.cpp:154-188: radius_/ratio_ are `unsigned int` there (:257-259), so 0.8 -> 0
  void configure(const ecto::tendrils& params, const ecto::tendrils&, const ecto::tendrils&) {
#ifdef TOD_AMD_WITH_ORK
    configure_impl();
#endif
    const std::string js = params.get<std::string>("search_json_params");
    radius_ = (unsigned int)json_number(js, "radius", 0);
    ratio_ = (unsigned int)json_number(js, "ratio", 0);
    // "ratio": 0.8 of the shipped configs is stored in an `unsigned int` by the reference, i.e. 0: its (empty) ratio block
    // never runs (:223-227). The test itself exists here (todhip_set_ratio_test) and is reachable through a key of its own,
    // "lowe_ratio" (a float in (0, 1]), so that the reference's configs keep their reference behaviour.
    lowe_ratio_ = (float)json_number(js, "lowe_ratio", 0.0);
    // The reference accepts only "LSH" and `throw;`s otherwise (:182-186). The LSH table parameters
    // (n_tables, key_size, multi_probe_level) select an *approximate* index there; here the search is exact.
    if (json_string(js, "type", "") != "LSH") throw std::runtime_error("Search not implemented for that type");
    if (!ctx_ && todhip_create(context_device(), nullptr, &ctx_) != TODHIP_OK)          // this cell's own: DB + search settings
      throw std::runtime_error("todhip_create failed: no usable MI355X / HIP device");
    if (todhip_set_ratio_test(ctx_, lowe_ratio_) != TODHIP_OK) throw std::runtime_error("lowe_ratio must lie in [0, 1]");
    // "approximate": 1 (a key of this adapter's own) makes those three parameters mean what they mean to the reference: an
    // LSH index of n_tables x key_size-bit keys probed multi_probe_level bits deep (todhip_set_lsh; FLANN's scheme, own key bits)
    if (json_number(js, "approximate", 0) != 0) {
      if (todhip_set_lsh(ctx_, (uint32_t)json_number(js, "n_tables", 0), (uint32_t)json_number(js, "key_size", 0),
                         (uint32_t)json_number(js, "multi_probe_level", 0)) != TODHIP_OK)
        throw std::runtime_error("LSH parameters out of range (n_tables <= 32, key_size <= 24, multi_probe_level <= 3)");
    } else if (todhip_set_lsh(ctx_, 0, 0, 0) != TODHIP_OK) {
      throw std::runtime_error("todhip_set_lsh failed");
    }
  }
  // DescriptorMatcher.cpp:60-129. `docs`: per object the "descriptors" (n x 32 CV_8U) and "points" attachments.
  struct ObjectModel { ObjectId id; cv::Mat descriptors, points; };
  void load_models(const std::vector<ObjectModel>& docs) {
    object_ids_.clear();
    std::vector<todhip_object> objs;
    keep_.clear();
    for (size_t i = 0; i < docs.size(); ++i) {
      cv::Mat pts = docs[i].points;
      if (pts.rows != 1) pts = pts.t();                                  // :84-85
      keep_.push_back(docs[i].descriptors.isContinuous() ? docs[i].descriptors : docs[i].descriptors.clone());
      keep_.push_back(pts.isContinuous() ? pts : pts.clone());
      todhip_object o;
      o.n = (uint32_t)keep_[2 * i].rows;
      o.desc = o.n ? keep_[2 * i].template ptr<uint8_t>(0) : nullptr;    // an empty model: no row 0 to point at
      o.pts_xyz = o.n ? keep_[2 * i + 1].template ptr<float>(0) : nullptr;
      objs.push_back(o);
      object_ids_.push_back(docs[i].id);
    }
    std::vector<float> spans(docs.size());
    const int rc = todhip_db_load(ctx_, objs.data(), (uint32_t)objs.size(), 32, 0, 1, spans.data());
    if (rc != TODHIP_OK) throw std::runtime_error("todhip_db_load failed");
    spans_.clear();
    for (size_t i = 0; i < docs.size(); ++i) spans_[docs[i].id] = spans[i];
    keep_.clear();
  }
#ifdef TOD_AMD_WITH_ORK
  void parameter_callback(const object_recognition_core::db::Documents& db_documents) {
    std::vector<ObjectModel> docs;
    BOOST_FOREACH(const object_recognition_core::db::Document& document, db_documents) {
      ObjectModel m;
      m.id = document.get_field<std::string>("object_id");
      document.get_attachment<cv::Mat>("descriptors", m.descriptors);
      document.get_attachment<cv::Mat>("points", m.points);
      docs.push_back(m);
    }
    load_models(docs);
  }
#endif
  // DescriptorMatcher.cpp:195-252
  int process(const ecto::tendrils& inputs, const ecto::tendrils& outputs) {
    const cv::Mat& descriptors = inputs.get<cv::Mat>("descriptors");
    std::vector<std::vector<cv::DMatch> > matches;
    const uint32_t nq = (uint32_t)descriptors.rows, k = 5;               // knnMatch(descriptors, matches, 5), :211
    std::vector<uint32_t> row_ptr(nq + 1, 0u);
    std::vector<todhip_dmatch> flat((size_t)nq * k);
    std::vector<float> xyz((size_t)nq * k * 3);
    if (radius_) {                                                       // :202
      cv::Mat q = descriptors.isContinuous() ? descriptors : descriptors.clone();
      const int rc = todhip_match(ctx_, q.template ptr<uint8_t>(0), nq, k, radius_, row_ptr.data(), flat.data(), xyz.data());
      if (rc == TODHIP_ENODB) return ecto::OK;                           // "No descriptors loaded", :204-208
      if (rc != TODHIP_OK) throw std::runtime_error("todhip_match failed");
      matches.resize(nq);
    }
    std::vector<cv::Mat> matches_3d(nq);
    for (uint32_t qi = 0; qi < nq && !matches.empty(); ++qi) {
      const uint32_t lo = row_ptr[qi], hi = row_ptr[qi + 1];
      matches[qi].resize(hi - lo);
      static_assert(sizeof(cv::DMatch) == sizeof(todhip_dmatch), "cv::DMatch layout");
      if (hi > lo) std::memcpy(&matches[qi][0], &flat[lo], (hi - lo) * sizeof(todhip_dmatch));
      matches_3d[qi] = cv::Mat(1, (int)(hi - lo), CV_32FC3);
      if (hi > lo) std::memcpy(matches_3d[qi].template ptr<float>(0), &xyz[3 * (size_t)lo], (hi - lo) * 12);
    }
    outputs["matches"] << matches;
    outputs["matches_3d"] << matches_3d;
    outputs["object_ids"] << object_ids_;
    outputs["spans"] << spans_;
    return ecto::OK;
  }

  DescriptorMatcher() {}
  ~DescriptorMatcher() { if (ctx_) todhip_destroy(ctx_); }
  DescriptorMatcher(const DescriptorMatcher&) = delete;                    // (ecto holds a cell's implementation by pointer)
  DescriptorMatcher& operator=(const DescriptorMatcher&) = delete;

  todhip_ctx* ctx_ = nullptr;
  unsigned int radius_ = 0, ratio_ = 0;
  float lowe_ratio_ = 0.f;
  std::vector<ObjectId> object_ids_;
  std::map<ObjectId, float> spans_;
  std::vector<cv::Mat> keep_;
};

// ------------------------------------------------------------------------------------------------------------
// What GuessGenerator publishes per pose when ORK-core's PoseResult is not available (the test double).
struct PoseOut { ObjectId object_id; float R[9]; float T[3]; std::vector<unsigned int> inlier_keypoints; };

struct GuessGenerator {
  // GuessGenerator.cpp:71-81
  static void declare_params(ecto::tendrils& params) {
    params.declare<unsigned int>("min_inliers", "Minimum number of inliers", 15u);
    params.declare<unsigned int>("n_ransac_iterations", "Number of RANSAC iterations.", 1000u);
    params.declare<float>("sensor_error", "The error (in meters) from the Kinect", 0.01f);
    params.declare<bool>("visualize", "If true, display temporary info through highgui", false);
    params.declare<std::string>("db", "The DB to get data from, as a JSON string").required(true);
    // extension, only read on the 2D-only branch (points3d empty and K given): the reprojection threshold of todhip_verify_2d
    params.declare<float>("reprojection_error", "2D-only branch: reprojection threshold in pixels", 3.0f);
  }
  // GuessGenerator.cpp:83-99
  static void declare_io(const ecto::tendrils&, ecto::tendrils& inputs, ecto::tendrils& outputs) {
    inputs.declare<cv::Mat>("image", "The height by width 3 channel point cloud");
    inputs.declare<cv::Mat>("points3d", "The height by width 3 channel point cloud");
    inputs.declare<std::vector<cv::KeyPoint> >("keypoints", "The interesting keypoints");
    inputs.declare<std::vector<std::vector<cv::DMatch> > >("matches", "The list of OpenCV DMatch");
    inputs.declare<std::vector<cv::Mat> >("matches_3d", "The corresponding 3d position of those matches. For each "
                                                        "point, a 1 by n 3 channel matrix (for x,y and z)");
    inputs.declare<std::map<ObjectId, float> >("spans", "For each found object, its span based on known features.");
    inputs.declare<std::vector<ObjectId> >("object_ids", "The ids used in the matches");
    // extension: with an empty points3d the reference does nothing (:147-152 "Only use 2d to 3d matching // TODO"); given the camera
    // matrix the cell solves that PnP problem with todhip_verify_2d instead. Left unconnected, the branch stays empty as in the reference.
    inputs.declare<cv::Mat>("K", "The camera matrix (3 x 3, float or double); only used when points3d is empty", cv::Mat());
#ifdef TOD_AMD_WITH_ORK
    outputs.declare<std::vector<object_recognition_core::common::PoseResult> >("pose_results", "The results of object recognition");
#else
    outputs.declare<std::vector<PoseOut> >("pose_results", "The results of object recognition");
#endif
    outputs.declare<std::vector<cv::Mat> >("Rs", "The rotations of the poses (useful for visualization)");
    outputs.declare<std::vector<cv::Mat> >("Ts", "The translations of the poses (useful for visualization)");
  }
  // GuessGenerator.cpp:101-120 (the visualisation colours are dropped, decision D5)
  void configure(const ecto::tendrils& params, const ecto::tendrils&, const ecto::tendrils&) {
    min_inliers_ = params.get<unsigned int>("min_inliers");
    n_ransac_iterations_ = params.get<unsigned int>("n_ransac_iterations");
    sensor_error_ = params.get<float>("sensor_error");
    reprojection_error_ = params.get<float>("reprojection_error");
#ifdef TOD_AMD_WITH_ORK
    db_ = object_recognition_core::db::ObjectDbParameters(params.get<std::string>("db")).generateDb();
#endif
    ctx_ = shared_context();
    todhip_rng_seed(&rng_, 1);          // the reference never calls srand (sac.h:71): one stream per process
  }
  // GuessGenerator.cpp:127-250
  int process(const ecto::tendrils& inputs, const ecto::tendrils& outputs) {
    const std::vector<std::vector<cv::DMatch> >& matches = inputs.get<std::vector<std::vector<cv::DMatch> > >("matches");
    const std::vector<cv::Mat>& matches_3d = inputs.get<std::vector<cv::Mat> >("matches_3d");
    const std::vector<cv::KeyPoint>& keypoints = inputs.get<std::vector<cv::KeyPoint> >("keypoints");
    const cv::Mat point_cloud = inputs.get<cv::Mat>("points3d");
    const std::vector<ObjectId>& object_ids_in = inputs.get<std::vector<ObjectId> >("object_ids");
    const std::map<ObjectId, float>& spans = inputs.get<std::map<ObjectId, float> >("spans");
    std::vector<cv::Mat> Rs, Ts;
#ifdef TOD_AMD_WITH_ORK
    std::vector<object_recognition_core::common::PoseResult> pose_results;
#else
    std::vector<PoseOut> pose_results;
#endif
    const cv::Mat Kmat = inputs.get<cv::Mat>("K");
    const bool with_cloud = !point_cloud.empty();
    if (with_cloud || !Kmat.empty()) {                                   // :147-152: without a cloud AND without K, nothing (the reference's TODO)
      const uint32_t nq = (uint32_t)matches.size();
      if (keypoints.size() < matches.size())                               // the library reads nq keypoints
        throw std::runtime_error("GuessGenerator: fewer keypoints than match lists");
      std::vector<float> kp(2 * (size_t)keypoints.size());
      for (size_t i = 0; i < keypoints.size(); ++i) { kp[2 * i] = keypoints[i].pt.x; kp[2 * i + 1] = keypoints[i].pt.y; }
      std::vector<uint32_t> row_ptr(nq + 1, 0u);
      for (uint32_t q = 0; q < nq; ++q) row_ptr[q + 1] = row_ptr[q] + (uint32_t)matches[q].size();
      std::vector<todhip_dmatch> flat(row_ptr[nq]);
      std::vector<float> xyz(3 * (size_t)row_ptr[nq]);
      for (uint32_t q = 0; q < nq; ++q) {
        const uint32_t n = (uint32_t)matches[q].size();
        if (!n) continue;
        std::memcpy(&flat[row_ptr[q]], &matches[q][0], n * sizeof(todhip_dmatch));
        std::memcpy(&xyz[3 * (size_t)row_ptr[q]], matches_3d[q].template ptr<float>(0), n * 12);
      }
      std::vector<float> span_by_index(object_ids_in.size(), 0.f);
      for (size_t o = 0; o < object_ids_in.size(); ++o) {
        std::map<ObjectId, float>::const_iterator it = spans.find(object_ids_in[o]);   // :186
        if (it != spans.end()) span_by_index[o] = it->second;
      }
      cv::Mat cloud;
      float K9[9] = {0};
      if (with_cloud) {
        cloud = point_cloud.isContinuous() ? point_cloud : point_cloud.clone();
      } else {
        if (Kmat.rows != 3 || Kmat.cols != 3 || (Kmat.type() != CV_32F && Kmat.type() != CV_64F))
          throw std::runtime_error("GuessGenerator: K must be a 3 x 3 float or double matrix");
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j)
            K9[3 * i + j] = Kmat.type() == CV_32F ? Kmat.template ptr<float>(i)[j] : (float)Kmat.template ptr<double>(i)[j];
      }
      todhip_verify_params prm = {min_inliers_, n_ransac_iterations_, with_cloud ? sensor_error_ : reprojection_error_};
      // the reference has no limit on the number of poses: on TODHIP_ECAPACITY the frame is redone from the same
      // generator state with twice the room
      std::vector<todhip_pose> poses;
      std::vector<uint32_t> inl;
      uint32_t n_poses = 0, n_inl = 0;
      int rc = TODHIP_ECAPACITY;
      for (size_t cap = 256; rc == TODHIP_ECAPACITY && cap <= (1u << 16); cap *= 2) {
        poses.resize(cap);
        inl.resize(cap * (keypoints.size() + 1));
        n_poses = (uint32_t)poses.size(); n_inl = (uint32_t)inl.size();
        todhip_rng rng = rng_;
        if (with_cloud)
          rc = todhip_verify(ctx_, kp.data(), nq, cloud.template ptr<float>(0), (uint32_t)cloud.rows, (uint32_t)cloud.cols,
                             row_ptr.data(), flat.data(), xyz.data(), span_by_index.data(), (uint32_t)span_by_index.size(), &prm,
                             &rng, poses.data(), &n_poses, inl.data(), &n_inl);
        else
          rc = todhip_verify_2d(ctx_, kp.data(), nq, K9, row_ptr.data(), flat.data(), xyz.data(), span_by_index.data(),
                                (uint32_t)span_by_index.size(), &prm, &rng, poses.data(), &n_poses, inl.data(), &n_inl);
        if (rc == TODHIP_OK) rng_ = rng;
      }
      if (rc != TODHIP_OK) throw std::runtime_error("todhip_verify failed");
      for (uint32_t i = 0; i < n_poses; ++i) {                            // :223-230
        cv::Mat R(3, 3, CV_32F), T(3, 1, CV_32F);
        std::memcpy(R.template ptr<float>(0), poses[i].R, 36);
        std::memcpy(T.template ptr<float>(0), poses[i].t, 12);
#ifdef TOD_AMD_WITH_ORK
        object_recognition_core::common::PoseResult pr;
        pr.set_R(R); pr.set_T(T); pr.set_object_id(db_, object_ids_in[poses[i].object]);
        pose_results.push_back(pr);
#else
        PoseOut pr;
        pr.object_id = object_ids_in[poses[i].object];
        std::memcpy(pr.R, poses[i].R, 36); std::memcpy(pr.T, poses[i].t, 12);
        pr.inlier_keypoints.assign(inl.begin() + poses[i].inlier_begin, inl.begin() + poses[i].inlier_end);
        pose_results.push_back(pr);
#endif
        Rs.push_back(R); Ts.push_back(T);
      }
    }
    outputs["pose_results"] << pose_results;
    outputs["Rs"] << Rs;
    outputs["Ts"] << Ts;
    return ecto::OK;
  }

  todhip_ctx* ctx_ = nullptr;
  todhip_rng rng_;
  unsigned int min_inliers_ = 15, n_ransac_iterations_ = 1000;
  float sensor_error_ = 0.01f, reprojection_error_ = 3.0f;
#ifdef TOD_AMD_WITH_ORK
  object_recognition_core::db::ObjectDbPtr db_;
#endif
};

// ------------------------------------------------------------------------------------------------------------
// Stage A. The reference's graph takes its features from ecto_opencv's FeatureDescriptor cell (third party; not in the
// reference tree): parameters json_feature_params / json_descriptor_params (the `feature:` / `descriptor:` blocks of
// conf/detection.ork:23-31 as JSON strings, forwarded at detector.py:35-36), inputs image and mask (detector.py:41,71-74),
// outputs keypoints and descriptors (detector.py:44,80-81). Only ORB exists here (the configs' type); `pattern` is an
// extension: the 256 x 4 int8 test pairs (x0, y0, x1, y1) -- OpenCV's learned bit_pattern_31_ when descriptors must
// interoperate with OpenCV-trained models; empty = the library's built-in seeded pattern.
struct FeatureDescriptor {
  static void declare_params(ecto::tendrils& p) {
    p.declare<std::string>("json_feature_params", "Parameters for the feature as a JSON string. It should have the format: "
                           "\"{\"type\":\"ORB/SIFT whatever\", \"module\":\"where_it_is\", \"param_1\":val1, ....}",
                           std::string("{\"type\": \"ORB\", \"module\": \"ecto_opencv.features2d\"}"));
    p.declare<std::string>("json_descriptor_params", "Parameters for the descriptor as a JSON string. It should have the "
                           "format: \"{\"type\":\"ORB/SIFT whatever\", \"module\":\"where_it_is\", \"param_1\":val1, ....}",
                           std::string("{\"type\": \"ORB\", \"module\": \"ecto_opencv.features2d\"}"));
    p.declare<cv::Mat>("pattern", "Optional 256 x 4 int8 rBRIEF test pairs (x0, y0, x1, y1); empty = built-in pattern", cv::Mat());
  }
  static void declare_io(const ecto::tendrils&, ecto::tendrils& inputs, ecto::tendrils& outputs) {
    inputs.declare<cv::Mat>("image", "An input image.");
    inputs.declare<cv::Mat>("mask", "An mask, same size as image.");
    outputs.declare<std::vector<cv::KeyPoint> >("keypoints", "The keypoints.");
    outputs.declare<cv::Mat>("descriptors", "The descriptors per keypoints");
  }
  void configure(const ecto::tendrils& params, const ecto::tendrils&, const ecto::tendrils&) {
    const std::string jf = params.get<std::string>("json_feature_params"), jd = params.get<std::string>("json_descriptor_params");
    if (json_string(jf, "type", "ORB") != "ORB" || json_string(jd, "type", "ORB") != "ORB")
      throw std::runtime_error("FeatureDescriptor: only ORB features/descriptors are implemented");
    n_features_ = (uint32_t)json_number(jf, "n_features", 1000);
    n_levels_ = (uint32_t)json_number(jf, "n_levels", 3);
    scale_factor_ = (float)json_number(jf, "scale_factor", 1.2);
    const cv::Mat& pat = params.get<cv::Mat>("pattern");
    pattern_.clear();
    if (!pat.empty()) {
      if (pat.rows * pat.cols != 1024) throw std::runtime_error("FeatureDescriptor: pattern must hold 256 x 4 int8 values");
      pattern_.assign(pat.template ptr<int8_t>(0), pat.template ptr<int8_t>(0) + 1024);
    }
    ctx_ = shared_context();
  }
  int process(const ecto::tendrils& inputs, const ecto::tendrils& outputs) {
    cv::Mat image = inputs.get<cv::Mat>("image");
    // cv::ORB converts a colour image itself (cvtColor(image, gray, CV_BGR2GRAY), third-party, recalled) and asserts 8-bit input;
    // here 3- and 4-channel 8-bit images are converted with cvtColor's fixed-point weights, anything else is refused -- never read
    // as if it were gray bytes
    if (image.type() == CV_8UC3 || image.type() == CV_8UC4) image = bgr_to_gray(image);
    if (!image.empty() && image.type() != CV_8UC1) throw std::runtime_error("FeatureDescriptor: the image must be 8-bit with 1, 3 or 4 channels");
    const cv::Mat& mask = inputs.get<cv::Mat>("mask");
    if (!image.isContinuous()) image = image.clone();
    const uint32_t H = (uint32_t)image.rows, W = (uint32_t)image.cols;
    if (!mask.empty() && ((uint32_t)mask.rows != H || (uint32_t)mask.cols != W || mask.type() != CV_8UC1))
      throw std::runtime_error("FeatureDescriptor: the mask must be 8-bit, one channel, of the image's size");
    std::vector<float> kp(2 * (size_t)n_features_), aux(4 * (size_t)n_features_);
    cv::Mat desc((int)n_features_, 32, CV_8U);
    uint32_t n = n_features_;
    const bool use_mask = !mask.empty() && (uint32_t)mask.rows == H && (uint32_t)mask.cols == W;
    cv::Mat mk = use_mask ? (mask.isContinuous() ? mask : mask.clone()) : cv::Mat();
    const int rc = todhip_orb_masked(ctx_, image.template ptr<uint8_t>(0), use_mask ? mk.template ptr<uint8_t>(0) : nullptr, H, W, W,
                                     n_features_, n_levels_, scale_factor_, pattern_.empty() ? nullptr : pattern_.data(), kp.data(),
                                     aux.data(), n_features_ ? desc.template ptr<uint8_t>(0) : nullptr, &n);
    if (rc != TODHIP_OK) throw std::runtime_error("todhip_orb failed");
    std::vector<cv::KeyPoint> keypoints(n);
    for (uint32_t i = 0; i < n; ++i) {
      keypoints[i].pt.x = kp[2 * i]; keypoints[i].pt.y = kp[2 * i + 1];
      keypoints[i].size = aux[4 * i]; keypoints[i].angle = aux[4 * i + 1]; keypoints[i].response = aux[4 * i + 2];
      keypoints[i].octave = (int)aux[4 * i + 3]; keypoints[i].class_id = -1;
    }
    cv::Mat out((int)n, 32, CV_8U);
    if (n) std::memcpy(out.template ptr<uint8_t>(0), desc.template ptr<uint8_t>(0), (size_t)n * 32);
    outputs["keypoints"] << keypoints;
    outputs["descriptors"] << out;
    return ecto::OK;
  }

  // Y = (1868 B + 9617 G + 4899 R + 2^13) >> 14: cvtColor's 8-bit BGR2GRAY / BGRA2GRAY
  static cv::Mat bgr_to_gray(const cv::Mat& src) {
    const int ch = src.type() == CV_8UC4 ? 4 : 3;
    cv::Mat g(src.rows, src.cols, CV_8UC1);
    for (int r = 0; r < src.rows; ++r) {
      const uint8_t* p = src.template ptr<uint8_t>(r);
      uint8_t* o = g.template ptr<uint8_t>(r);
      for (int c = 0; c < src.cols; ++c, p += ch) o[c] = (uint8_t)((1868u * p[0] + 9617u * p[1] + 4899u * p[2] + 8192u) >> 14);
    }
    return g;
  }

  todhip_ctx* ctx_ = nullptr;
  uint32_t n_features_ = 1000, n_levels_ = 3;
  float scale_factor_ = 1.2f;
  std::vector<int8_t> pattern_;
};

}  // namespace tod_amd

#ifdef TOD_AMD_WITH_ORK
// src/detection/module.cpp:38 and the ECTO_CELL lines of the two reference files (:269, :275)
ECTO_CELL(ecto_detection, tod_amd::DescriptorMatcher, "DescriptorMatcher", "Given descriptors, find matches, relating to objects.");
ECTO_CELL(ecto_detection, tod_amd::GuessGenerator, "GuessGenerator", "Given descriptors and 3D positions, compute object guesses.");
// stage A: registered under the name detector.py imports from ecto_opencv.features2d (INTEGRATION.md shows the one-line
// change of that import, or the ecto_opencv-side shadowing that needs no change at all)
ECTO_CELL(ecto_detection, tod_amd::FeatureDescriptor, "FeatureDescriptor", "Compute features and descriptors for an image (ORB on the GPU).");
#endif
