// Drives adapter/ecto_cells.hpp the way the ecto scheduler and python/object_recognition_tod/detector.py:56-110
// do: declare_params / declare_io / configure / (model load) / process, DescriptorMatcher -> GuessGenerator.
// Inputs are read from binary files written by tests/test_adapter_gpu.py; outputs are written back for comparison.
#include "mini_ecto/mini_ecto.hpp"
#include "../adapter/ecto_cells.hpp"

#include <cstdio>
#include <fstream>
#include <iostream>

template <typename T> static std::vector<T> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot open " + path);
  const size_t n = (size_t)f.tellg();
  std::vector<T> v(n / sizeof(T));
  f.seekg(0);
  f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
  return v;
}
template <typename T> static void dump(const std::string& path, const std::vector<T>& v) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char** argv) {
  if (argc < 2) { std::cerr << "usage: adapter_test <dir> [declare-only]\n"; return 2; }
  const std::string dir = argv[1];
  try {
    // ---- what the cells declare must carry the reference's names
    ecto::tendrils mp, mi, mo, gp, gi, go;
    tod_amd::DescriptorMatcher::declare_params(mp);
    tod_amd::DescriptorMatcher::declare_io(mp, mi, mo);
    tod_amd::GuessGenerator::declare_params(gp);
    tod_amd::GuessGenerator::declare_io(gp, gi, go);
    const char* want_m_out[] = {"matches", "matches_3d", "object_ids", "spans"};
    const char* want_g_in[] = {"image", "points3d", "keypoints", "matches", "matches_3d", "spans", "object_ids"};
    const char* want_g_out[] = {"pose_results", "Rs", "Ts"};
    const char* want_g_par[] = {"min_inliers", "n_ransac_iterations", "sensor_error", "visualize", "db"};
    if (!mp.has("search_json_params") || !mi.has("descriptors")) return 3;
    for (const char* n : want_m_out) if (!mo.has(n)) return 3;
    for (const char* n : want_g_in) if (!gi.has(n)) return 3;
    for (const char* n : want_g_out) if (!go.has(n)) return 3;
    for (const char* n : want_g_par) if (!gp.has(n)) return 3;
    if (gp.get<unsigned int>("min_inliers") != 15u || gp.get<unsigned int>("n_ransac_iterations") != 1000u) return 3;
    // stage A: the names detector.py:35-36,41,44,71-74,80-81 uses on ecto_opencv's FeatureDescriptor
    ecto::tendrils fp, fi, fo;
    tod_amd::FeatureDescriptor::declare_params(fp);
    tod_amd::FeatureDescriptor::declare_io(fp, fi, fo);
    if (!fp.has("json_feature_params") || !fp.has("json_descriptor_params") || !fi.has("image") || !fi.has("mask") ||
        !fo.has("keypoints") || !fo.has("descriptors")) return 3;
    if (argc > 2) { std::cout << "declare ok\n"; return 0; }

    // ---- stage A as conf/detection.ork:23-31 configures it (n_features reduced to the frame's 500), with and without a mask
    {
      fp["json_feature_params"] << std::string("{\"type\": \"ORB\", \"module\": \"ecto_opencv.features2d\", \"n_features\": 500, "
                                               "\"n_levels\": 3, \"scale_factor\": 1.2}");
      fp["json_descriptor_params"] << std::string("{\"type\": \"ORB\", \"module\": \"ecto_opencv.features2d\"}");
      tod_amd::FeatureDescriptor feat;
      feat.configure(fp, fi, fo);
      std::vector<uint8_t> img = slurp<uint8_t>(dir + "/image.bin"), msk = slurp<uint8_t>(dir + "/mask.bin");
      cv::Mat im(480, 640, CV_8U), mk(480, 640, CV_8U);
      std::memcpy(im.ptr<uint8_t>(0), img.data(), img.size());
      std::memcpy(mk.ptr<uint8_t>(0), msk.data(), msk.size());
      for (int pass = 0; pass < 2; ++pass) {
        fi["image"] << im;
        fi["mask"] << (pass ? mk : cv::Mat());
        if (feat.process(fi, fo) != ecto::OK) return 7;
        const std::vector<cv::KeyPoint>& kps = fo.get<std::vector<cv::KeyPoint> >("keypoints");
        const cv::Mat& d = fo.get<cv::Mat>("descriptors");
        if ((int)kps.size() != d.rows || d.cols != 32) return 7;
        std::vector<float> kpf;
        for (const cv::KeyPoint& k : kps) { kpf.push_back(k.pt.x); kpf.push_back(k.pt.y); kpf.push_back(k.size); kpf.push_back(k.angle);
                                            kpf.push_back(k.response); kpf.push_back((float)k.octave); }
        std::vector<uint8_t> dv(d.ptr<uint8_t>(0), d.ptr<uint8_t>(0) + (size_t)d.rows * 32);
        dump(dir + (pass ? "/out_orb_kp_masked.bin" : "/out_orb_kp.bin"), kpf);
        dump(dir + (pass ? "/out_orb_desc_masked.bin" : "/out_orb_desc.bin"), dv);
      }
      // a BGR image whose three channels equal the gray one gives the gray image's keypoints (the cell converts as cv::ORB does);
      // a 16-bit image, and a mask of another size, are refused instead of being read as gray bytes
      {
        cv::Mat bgr(480, 640, CV_8UC3);
        for (size_t i = 0; i < img.size(); ++i) { uint8_t* p = bgr.ptr<uint8_t>(0) + 3 * i; p[0] = p[1] = p[2] = img[i]; }
        fi["image"] << bgr;
        fi["mask"] << cv::Mat();
        if (feat.process(fi, fo) != ecto::OK) return 7;
        const cv::Mat& d = fo.get<cv::Mat>("descriptors");
        std::vector<uint8_t> dv(d.ptr<uint8_t>(0), d.ptr<uint8_t>(0) + (size_t)d.rows * 32);
        dump(dir + "/out_orb_desc_bgr.bin", dv);
        bool threw = false;
        fi["image"] << cv::Mat(480, 640, CV_16U);
        try { feat.process(fi, fo); } catch (const std::exception&) { threw = true; }
        if (!threw) return 9;
        threw = false;
        fi["image"] << im;
        fi["mask"] << cv::Mat(240, 320, CV_8U);
        try { feat.process(fi, fo); } catch (const std::exception&) { threw = true; }
        if (!threw) return 9;
      }
      ecto::tendrils badp;
      tod_amd::FeatureDescriptor::declare_params(badp);
      badp["json_feature_params"] << std::string("{\"type\": \"SIFT\"}");
      bool threw = false;
      try { tod_amd::FeatureDescriptor f2; f2.configure(badp, fi, fo); } catch (const std::exception&) { threw = true; }
      if (!threw) return 8;
    }

    // ---- configure as conf/detection.ork:32-42 would
    mp["search_json_params"] << std::string("{\"type\": \"LSH\", \"module\": \"ecto_opencv.features2d\", \"key_size\": 16, "
                                            "\"multi_probe_level\": 1, \"n_tables\": 10, \"radius\": 35, \"ratio\": 0.8}");
    gp["min_inliers"] << 8u;
    gp["n_ransac_iterations"] << 2500u;
    gp["sensor_error"] << 0.01f;
    gp["db"] << std::string("{}");
    tod_amd::DescriptorMatcher matcher;
    tod_amd::GuessGenerator guess;
    matcher.configure(mp, mi, mo);
    guess.configure(gp, gi, go);
    // a second matcher cell with other search parameters (a ratio test, the approximate index) has a context of its own: it must
    // not change what the first one computes (its DB stays empty; the frame below goes through `matcher`)
    tod_amd::DescriptorMatcher other;
    {
      ecto::tendrils op, oi, oo;
      tod_amd::DescriptorMatcher::declare_params(op);
      tod_amd::DescriptorMatcher::declare_io(op, oi, oo);
      op["search_json_params"] << std::string("{\"type\": \"LSH\", \"radius\": 20, \"lowe_ratio\": 0.5, \"approximate\": 1, "
                                              "\"n_tables\": 4, \"key_size\": 12, \"multi_probe_level\": 1}");
      other.configure(op, oi, oo);
      if (other.ctx_ == matcher.ctx_ || !other.ctx_) return 10;
    }

    // ---- models (what parameter_callback receives from the DB)
    std::vector<uint32_t> off = slurp<uint32_t>(dir + "/obj_off.bin");
    std::vector<uint8_t> desc = slurp<uint8_t>(dir + "/desc.bin");
    std::vector<float> pts = slurp<float>(dir + "/pts.bin");
    std::vector<tod_amd::DescriptorMatcher::ObjectModel> docs;
    for (size_t o = 0; o + 1 < off.size(); ++o) {
      const int n = (int)(off[o + 1] - off[o]);
      tod_amd::DescriptorMatcher::ObjectModel m;
      m.id = "object_" + std::to_string(o);
      m.descriptors = cv::Mat(n, 32, CV_8U);
      m.points = cv::Mat(n, 1, CV_32FC3);                      // stored n x 1: the cell transposes to 1 x n (:84-85)
      if (n) {
        std::memcpy(m.descriptors.ptr<uint8_t>(0), &desc[(size_t)off[o] * 32], (size_t)n * 32);
        std::memcpy(m.points.ptr<float>(0), &pts[(size_t)off[o] * 3], (size_t)n * 12);
      }
      docs.push_back(m);
    }
    matcher.load_models(docs);

    // ---- one frame through both cells
    std::vector<uint8_t> q = slurp<uint8_t>(dir + "/q_desc.bin");
    std::vector<float> kp = slurp<float>(dir + "/kp_xy.bin");
    std::vector<float> cloud = slurp<float>(dir + "/cloud.bin");
    const int nq = (int)(q.size() / 32), H = 480, W = 640;
    cv::Mat qm(nq, 32, CV_8U);
    std::memcpy(qm.ptr<uint8_t>(0), q.data(), q.size());
    mi["descriptors"] << qm;
    if (matcher.process(mi, mo) != ecto::OK) return 4;

    std::vector<cv::KeyPoint> kps(nq);
    for (int i = 0; i < nq; ++i) { kps[i].pt.x = kp[2 * i]; kps[i].pt.y = kp[2 * i + 1]; }
    cv::Mat cm(H, W, CV_32FC3);
    std::memcpy(cm.ptr<float>(0), cloud.data(), cloud.size() * 4);
    gi["image"] << cv::Mat();
    gi["points3d"] << cm;
    gi["keypoints"] << kps;
    gi["matches"] << mo.get<std::vector<std::vector<cv::DMatch> > >("matches");
    gi["matches_3d"] << mo.get<std::vector<cv::Mat> >("matches_3d");
    gi["spans"] << mo.get<std::map<tod_amd::ObjectId, float> >("spans");
    gi["object_ids"] << mo.get<std::vector<tod_amd::ObjectId> >("object_ids");
    if (guess.process(gi, go) != ecto::OK) return 5;

    // ---- outputs for the Python side
    const std::vector<std::vector<cv::DMatch> >& matches = mo.get<std::vector<std::vector<cv::DMatch> > >("matches");
    std::vector<int32_t> flat;
    std::vector<float> dist;
    for (size_t qi = 0; qi < matches.size(); ++qi)
      for (const cv::DMatch& m : matches[qi]) { flat.push_back(m.queryIdx); flat.push_back(m.trainIdx); flat.push_back(m.imgIdx); dist.push_back(m.distance); }
    dump(dir + "/out_matches.bin", flat);
    dump(dir + "/out_dist.bin", dist);
    const std::vector<tod_amd::PoseOut>& poses = go.get<std::vector<tod_amd::PoseOut> >("pose_results");
    std::vector<float> rt;
    std::vector<uint32_t> inl;
    for (const tod_amd::PoseOut& p : poses) {
      for (float v : p.R) rt.push_back(v);
      for (float v : p.T) rt.push_back(v);
      inl.push_back((uint32_t)std::stoul(p.object_id.substr(7)));
      inl.push_back((uint32_t)p.inlier_keypoints.size());
      for (unsigned v : p.inlier_keypoints) inl.push_back(v);
    }
    dump(dir + "/out_poses.bin", rt);
    dump(dir + "/out_inliers.bin", inl);
    // ---- the same frame without a cloud: nothing, as in the reference (GuessGenerator.cpp:147-152) ...
    gi["points3d"] << cv::Mat();
    if (guess.process(gi, go) != ecto::OK) return 8;
    if (!go.get<std::vector<tod_amd::PoseOut> >("pose_results").empty()) return 8;
    // ... and with the camera matrix connected: the PnP extension (todhip_verify_2d)
    cv::Mat Km = cv::Mat(3, 3, CV_64F);
    const double Kv[9] = {525.0, 0, 320.0, 0, 525.0, 240.0, 0, 0, 1};
    std::memcpy(Km.ptr<double>(0), Kv, sizeof(Kv));
    gi["K"] << Km;
    if (guess.process(gi, go) != ecto::OK) return 8;
    {
      std::vector<float> rt2;
      std::vector<uint32_t> inl2;
      for (const tod_amd::PoseOut& p : go.get<std::vector<tod_amd::PoseOut> >("pose_results")) {
        for (float v : p.R) rt2.push_back(v);
        for (float v : p.T) rt2.push_back(v);
        inl2.push_back((uint32_t)std::stoul(p.object_id.substr(7)));
        inl2.push_back((uint32_t)p.inlier_keypoints.size());
        for (unsigned v : p.inlier_keypoints) inl2.push_back(v);
      }
      dump(dir + "/out_poses_2d.bin", rt2);
      dump(dir + "/out_inliers_2d.bin", inl2);
    }
    const std::vector<cv::Mat>& Rs = go.get<std::vector<cv::Mat> >("Rs");
    std::cout << "adapter ok: " << dist.size() << " matches, " << poses.size() << " poses, " << Rs.size() << " Rs\n";
    // a non-LSH search type must throw, as DescriptorMatcher.cpp:182-186 does
    ecto::tendrils bad;
    tod_amd::DescriptorMatcher::declare_params(bad);
    bad["search_json_params"] << std::string("{\"type\": \"KDTREE\", \"radius\": 35}");
    bool threw = false;
    try { tod_amd::DescriptorMatcher m2; m2.configure(bad, mi, mo); } catch (const std::exception&) { threw = true; }
    if (!threw) return 6;
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "adapter_test: " << e.what() << "\n";
    return 1;
  }
}
