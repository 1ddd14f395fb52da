// plugin_host — a minimal GSLAM host that exercises the drop-in boundary exactly as an application
// plugin would: GSLAM::Optimizer::create() (GSLAM/core/Optimizer.h:234-248) and
// GSLAM::FeatureDetector::create() resolve the plugins through GSLAM::Registry / dlopen, the host fills
// GSLAM's own containers (BundleGraph, GImage, KeyPoint) and calls the virtuals.  Used by
// tests/test_plugins_gpu.py; built here (needs the GSLAM headers), runs on the GPU box.
//
//   plugin_host ba   <plugin_dir> <graph.bin> <out.bin>
//   plugin_host pnp  <plugin_dir> <pnp.bin> <out.bin>
//   plugin_host pg   <plugin_dir> <posegraph.bin> <out.bin>    (optimize() on se3Graph / sim3Graph / gpsGraph edges)
//   plugin_host align <plugin_dir> <align.bin> <out.bin>       (optimizeICP, fitSim3, optimizePose)
//   plugin_host orb  <plugin_dir> <w> <h> <channels> <image.raw> <out.bin> <K>
//   plugin_host orbbatch <plugin_dir> <w> <h> <channels> <n> <frames.raw> <out.bin> <K>   (detectAndComputeBatch and the
//                    asynchronous submit / collect pair against per-frame detectAndCompute, all three written out)
//   plugin_host lat  <plugin_dir> <w> <h> <n> <frames.raw> <K> <iterations>   (single-frame latency of
//                    detectAndCompute, match, optimizePnP through the plugins: p50 / p99; async frames per second)
//   plugin_host bow  <plugin_dir> <vocab.gbow> <desc.raw> <n> <levelsup> <out.bin>
//   plugin_host undist <plugin_dir> <channels> <image.raw>     (fixed OpenCV-model camera 320x240 -> pinhole)
//   plugin_host est  <plugin_dir> <model 0|1|2> <n> <pts.raw (n x 4 doubles: src xy, dst xy)> <thr> <out.bin>
//   plugin_host app  <plugin_dir> <w> <h> <n_frames> <frames.raw> <out.bin> <K>   (launcher-style: loads the
//                    `orbhip` application plugin exactly as GSLAM/gslam/main.cpp:18-45 does and feeds "dataset/frame")
#include <GSLAM/core/GSLAM.h>
#include <GSLAM/core/Optimizer.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <iostream>
#include <vector>

#include <GSLAM/core/Vocabulary.h>

#include <GSLAM/core/Estimator.h>

#include "FeatureDetector.h"
#include "UndistorterHIP.h"

using namespace GSLAM;

template <typename T>
static std::vector<T> read_vec(std::ifstream& f, size_t n) {
  std::vector<T> v(n);
  if (n) f.read((char*)v.data(), n * sizeof(T));
  return v;
}

// Optimizer::magin through the plugin: same input file as run_ba; out = int32 ok, int32 n_edges, then per edge
// int32 first, int32 second, 7 doubles measurement [qx qy qz qw tx ty tz], 36 doubles information; then the converted
// graph (keyframes + SE3 edges, no observations) goes back through optimize() -- the pose-graph path of the same plugin.
// Self-calibrating bundle adjustment through Optimizer::optimize (BundleGraph::camera + cameraDOF, Optimizer.h:86-100,
// 169-171).  The host does what a GSLAM front end does: pixels -> CameraAnchors with the camera it believes in
// (camera.UnProject), the graph carries that camera and the parameters to free.  File: int32 {nc, np, no, camera dof,
// max iterations, n camera parameters}, double huber (on the z = 1 plane), camera parameters (w h fx fy cx cy [k1 k2 p1 p2 k3]),
// poses nc x 7, keyframe dof nc, points np x 3, observation frame / point no each, pixels no x 2.
static int run_calib(const std::string& dir, const char* in, const char* out) {
  svar.GetString("OptimizerPlugin", "") = dir + "/libgslam_optimizer.so";
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[6];
  f.read((char*)hdr, sizeof(hdr));
  double huber;
  f.read((char*)&huber, 8);
  const int nc = hdr[0], np = hdr[1], no = hdr[2];
  std::vector<double> cam = read_vec<double>(f, hdr[5]);
  std::vector<double> pose = read_vec<double>(f, (size_t)nc * 7);
  std::vector<int32_t> dof = read_vec<int32_t>(f, nc);
  std::vector<double> xyz = read_vec<double>(f, (size_t)np * 3);
  std::vector<int32_t> ocam = read_vec<int32_t>(f, no), opt = read_vec<int32_t>(f, no);
  std::vector<double> px = read_vec<double>(f, (size_t)no * 2);
  OptimizerPtr opt_ptr = Optimizer::create();
  if (!opt_ptr) { std::cerr << "Optimizer::create() returned null\n"; return 2; }
  opt_ptr->_config.projectErrorHuberThreshold = huber;
  opt_ptr->_config.maxIterations = hdr[4];
  BundleGraph g;
  g.camera = Camera(cam);
  g.cameraDOF = (CameraEstimationDOF)hdr[3];
  g.keyframes.resize(nc);
  for (int i = 0; i < nc; ++i) {
    const double* p = &pose[(size_t)i * 7];
    g.keyframes[i].estimation = SIM3(SO3(p[0], p[1], p[2], p[3]), Point3d(p[4], p[5], p[6]), 1.0);
    g.keyframes[i].dof = (KeyFrameEstimzationDOF)dof[i];
  }
  g.mappoints.resize(np);
  for (int i = 0; i < np; ++i) g.mappoints[i] = std::make_pair(Point3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), true);
  g.mappointObserves.resize(no);
  double worst_roundtrip = 0;
  for (int k = 0; k < no; ++k) {
    BundleEdge e;
    e.pointId = opt[k];
    e.frameId = ocam[k];
    e.measurement = g.camera.UnProject(Point2d(px[2 * k], px[2 * k + 1]));
    const Point2d back = g.camera.Project(e.measurement);
    worst_roundtrip = std::max(worst_roundtrip, std::max(std::fabs(back.x - px[2 * k]), std::fabs(back.y - px[2 * k + 1])));
    e.information = NULL;
    g.mappointObserves[k] = e;
  }
  const bool ok = opt_ptr->optimize(g);
  const std::vector<double> res = g.camera.getParameters();
  std::cout.precision(9);
  std::cout << "calib=" << ok << " camera=" << g.camera.info() << " unproject_project_roundtrip_px=" << worst_roundtrip << std::endl;
  std::ofstream o(out, std::ios::binary);
  int32_t okv = ok ? 1 : 0, n = (int32_t)res.size();
  o.write((char*)&okv, 4);
  o.write((char*)&n, 4);
  o.write((char*)res.data(), res.size() * 8);
  o.write((char*)&worst_roundtrip, 8);
  for (int i = 0; i < nc; ++i) {
    const SIM3& T = g.keyframes[i].estimation;
    const SO3 r = T.get_rotation();
    const Point3d t = T.get_translation();
    double m[7] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
    o.write((char*)m, sizeof(m));
  }
  for (int i = 0; i < np; ++i) {
    double m[3] = {g.mappoints[i].first.x, g.mappoints[i].first.y, g.mappoints[i].first.z};
    o.write((char*)m, sizeof(m));
  }
  return ok ? 0 : 3;
}

static int run_magin(const std::string& dir, const char* in, const char* out) {
  svar.GetString("OptimizerPlugin", "") = dir + "/libgslam_optimizer.so";
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[6];
  f.read((char*)hdr, sizeof(hdr));
  double huber;
  f.read((char*)&huber, 8);
  const int nc = hdr[0], np = hdr[1], no = hdr[2];
  std::vector<double> pose = read_vec<double>(f, (size_t)nc * 7);
  std::vector<int32_t> dof = read_vec<int32_t>(f, nc);
  std::vector<double> xyz = read_vec<double>(f, (size_t)np * 3);
  std::vector<uint8_t> pfree = read_vec<uint8_t>(f, hdr[5] ? np : 0);
  std::vector<int32_t> ocam = read_vec<int32_t>(f, no), opt = read_vec<int32_t>(f, no);
  std::vector<double> oxy = read_vec<double>(f, (size_t)no * 2);
  std::vector<double> info = read_vec<double>(f, hdr[3] ? (size_t)no * 4 : 0);
  OptimizerPtr opt_ptr = Optimizer::create();
  if (!opt_ptr) { std::cerr << "Optimizer::create() returned null\n"; return 2; }
  opt_ptr->_config.projectErrorHuberThreshold = huber;
  opt_ptr->_config.maxIterations = hdr[4];
  BundleGraph g;
  g.cameraDOF = UPDATE_CAMERA_NONE;
  g.keyframes.resize(nc);
  for (int i = 0; i < nc; ++i) {
    const double* p = &pose[(size_t)i * 7];
    g.keyframes[i].estimation = SIM3(SO3(p[0], p[1], p[2], p[3]), Point3d(p[4], p[5], p[6]), 1.0);
    g.keyframes[i].dof = (KeyFrameEstimzationDOF)dof[i];
  }
  g.mappoints.resize(np);
  for (int i = 0; i < np; ++i)
    g.mappoints[i] = std::make_pair(Point3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), hdr[5] ? pfree[i] != 0 : true);
  g.mappointObserves.resize(no);
  for (int k = 0; k < no; ++k) {
    BundleEdge e;
    e.pointId = opt[k];
    e.frameId = ocam[k];
    e.measurement = Point3d(oxy[2 * k], oxy[2 * k + 1], 1.0);
    e.information = hdr[3] ? &info[(size_t)k * 4] : NULL;
    g.mappointObserves[k] = e;
  }
  const bool ok = opt_ptr->magin(g);
  std::cout << "magin=" << ok << " edges=" << g.se3Graph.size() << " observations_left=" << g.mappointObserves.size() << std::endl;
  std::ofstream o(out, std::ios::binary);
  int32_t okv = ok ? 1 : 0, ne = (int32_t)g.se3Graph.size();
  o.write((char*)&okv, 4);
  o.write((char*)&ne, 4);
  for (size_t k = 0; k < g.se3Graph.size(); ++k) {
    const SE3Edge& e = g.se3Graph[k];
    int32_t ij[2] = {(int32_t)e.firstId, (int32_t)e.secondId};
    o.write((char*)ij, 8);
    const SO3 r = e.measurement.get_rotation();
    const Point3d t = e.measurement.get_translation();
    double m[7] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
    o.write((char*)m, sizeof(m));
    o.write((char*)e.information, 36 * sizeof(double));
  }
  if (ok) {
    // the result is a pose graph the same plugin solves: start from perturbed keyframes, the first one fixed
    BundleGraph pg = g;
    for (size_t i = 1; i < pg.keyframes.size(); ++i) {
      SIM3& T = pg.keyframes[i].estimation;
      T = SIM3(T.get_rotation(), T.get_translation() + Point3d(0.01 * (i % 3), -0.01 * (i % 2), 0.005), 1.0);
    }
    pg.keyframes[0].dof = UPDATE_KF_NONE;
    const bool okp = opt_ptr->optimize(pg);
    double worst = 0;
    for (size_t k = 0; k < pg.se3Graph.size(); ++k) {
      const SE3Edge& e = pg.se3Graph[k];
      const SIM3 &Ti = pg.keyframes[e.firstId].estimation, &Tj = pg.keyframes[e.secondId].estimation;
      const SE3 rel = SE3(Ti.get_rotation(), Ti.get_translation()).inverse() * SE3(Tj.get_rotation(), Tj.get_translation());
      const SE3 err = e.measurement.inverse() * rel;
      const SO3 q = err.get_rotation();
      worst = std::max(worst, std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z));
    }
    std::cout.precision(6);
    std::cout << "posegraph_optimize=" << okp << " worst_rotation_residual=" << worst << std::endl;
  }
  return 0;
}

static int run_ba(const std::string& dir, const char* in, const char* out, double kf_scale, int zero_z_obs) {
  svar.GetString("OptimizerPlugin", "") = dir + "/libgslam_optimizer.so";
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[6];  // nc, np, no, has_info, max_iterations, has_pfree
  f.read((char*)hdr, sizeof(hdr));
  double huber;
  f.read((char*)&huber, 8);
  const int nc = hdr[0], np = hdr[1], no = hdr[2];
  std::vector<double> pose = read_vec<double>(f, (size_t)nc * 7);
  std::vector<int32_t> dof = read_vec<int32_t>(f, nc);
  std::vector<double> xyz = read_vec<double>(f, (size_t)np * 3);
  std::vector<uint8_t> pfree = read_vec<uint8_t>(f, hdr[5] ? np : 0);
  std::vector<int32_t> ocam = read_vec<int32_t>(f, no), opt = read_vec<int32_t>(f, no);
  std::vector<double> oxy = read_vec<double>(f, (size_t)no * 2);
  std::vector<double> info = read_vec<double>(f, hdr[3] ? (size_t)no * 4 : 0);

  OptimizerPtr opt_ptr = Optimizer::create();
  if (!opt_ptr) { std::cerr << "Optimizer::create() returned null\n"; return 2; }
  opt_ptr->_config.projectErrorHuberThreshold = huber;
  opt_ptr->_config.maxIterations = hdr[4];
  BundleGraph g;
  g.cameraDOF = UPDATE_CAMERA_NONE;
  g.keyframes.resize(nc);
  for (int i = 0; i < nc; ++i) {
    const double* p = &pose[(size_t)i * 7];
    g.keyframes[i].estimation = SIM3(SO3(p[0], p[1], p[2], p[3]), Point3d(p[4], p[5], p[6]), kf_scale);
    g.keyframes[i].dof = (KeyFrameEstimzationDOF)dof[i];
  }
  g.mappoints.resize(np);
  for (int i = 0; i < np; ++i)
    g.mappoints[i] = std::make_pair(Point3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), hdr[5] ? pfree[i] != 0 : true);
  g.mappointObserves.resize(no);
  for (int k = 0; k < no; ++k) {
    BundleEdge e;
    e.pointId = opt[k];
    e.frameId = ocam[k];
    e.measurement = Point3d(oxy[2 * k], oxy[2 * k + 1], 1.0);
    e.information = hdr[3] ? &info[(size_t)k * 4] : NULL;
    g.mappointObserves[k] = e;
  }
  if (zero_z_obs >= 0 && zero_z_obs < no) g.mappointObserves[zero_z_obs].measurement.z = 0;  // a caller bug: must fail
  if (getenv("GSLAM_HOST_WARM_GRAPH")) {
    // a back end re-optimises the same window: a first call on a copy with other values leaves the plugin's resident
    // graph behind, the checked call below then goes through the update path (same topology, new values)
    BundleGraph warm = g;
    for (size_t i = 0; i < warm.mappoints.size(); ++i) warm.mappoints[i].first = warm.mappoints[i].first + Point3d(0.01, -0.02, 0.015);
    const bool okw = opt_ptr->optimize(warm);
    std::cout << "warm_optimize=" << okw << std::endl;
  }
  const bool ok = opt_ptr->optimize(g);
  // sum of squared reprojection errors at the result through the reference's OWN SIM3 algebra (scale included):
  // X_c = T_wc^-1 X_w  (GSLAM/core/SIM3.h:120-131)
  double ssq = 0, min_scale = 1e300, max_scale = 0;
  for (int k = 0; k < no; ++k) {
    const BundleEdge& e = g.mappointObserves[k];
    if (e.measurement.z == 0) continue;
    // (SIM3::inv() itself does not compile in this snapshot -- it calls a non-existent SO3::inv(), SIM3.h:126-131 -- so
    // the inverse of  X_w = R (s X_c) + t  (SIM3.h:120-123) is spelled out with the reference's SO3 operators)
    const SIM3& T = g.keyframes[e.frameId].estimation;
    const Point3d Xc = (T.get_rotation().inverse() * (g.mappoints[e.pointId].first - T.get_translation())) * (1.0 / T.get_scale());
    const Point3d back = T * Xc;  // the reference's forward map must return the world point
    if ((back - g.mappoints[e.pointId].first).norm() > 1e-9) { std::cerr << "SIM3 forward/inverse mismatch\n"; return 4; }
    const double rx = Xc.x / Xc.z - e.measurement.x / e.measurement.z, ry = Xc.y / Xc.z - e.measurement.y / e.measurement.z;
    ssq += rx * rx + ry * ry;
  }
  for (int i = 0; i < nc; ++i) {
    min_scale = std::min(min_scale, g.keyframes[i].estimation.get_scale());
    max_scale = std::max(max_scale, g.keyframes[i].estimation.get_scale());
  }
  std::cout.precision(17);
  std::cout << "ref_sim3_ssq=" << ssq << " scale_min=" << min_scale << " scale_max=" << max_scale << std::endl;
  std::ofstream o(out, std::ios::binary);
  int32_t okv = ok ? 1 : 0;
  o.write((char*)&okv, 4);
  for (int i = 0; i < nc; ++i) {
    const SIM3& T = g.keyframes[i].estimation;
    SO3 r = T.get_rotation();
    Point3d t = T.get_translation();
    double p[7] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
    o.write((char*)p, sizeof(p));
  }
  for (int i = 0; i < np; ++i) {
    double p[3] = {g.mappoints[i].first.x, g.mappoints[i].first.y, g.mappoints[i].first.z};
    o.write((char*)p, sizeof(p));
  }
  // the other virtuals keep the interface's "unsupported" default
  std::vector<std::pair<CameraAnchor, CameraAnchor> > m;
  std::vector<IdepthEstimation> id;
  SE3 rel;
  std::cout << "optimize=" << ok << " optimizePose(unsupported)=" << opt_ptr->optimizePose(m, id, rel) << std::endl;  // (no matches)
  return ok ? 0 : 3;
}

// Pose graph through Optimizer::optimize: keyframes + se3Graph / sim3Graph / gpsGraph (Optimizer.h:127-148).
// File: int32 {nf, n_se3, n_sim3, n_gps, has_info, max_iterations}; frames nf x 8 [qx qy qz qw tx ty tz s]; dof nf;
// se3: first, second, meas n x 7 [, info n x 36]; sim3: first, second, meas n x 8 [, info n x 49]; gps: frame, meas n x 7 [, info].
static int run_pg(const std::string& dir, const char* in, const char* out) {
  svar.GetString("OptimizerPlugin", "") = dir + "/libgslam_optimizer.so";
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[6];
  f.read((char*)hdr, sizeof(hdr));
  const int nf = hdr[0], n1 = hdr[1], n2 = hdr[2], ng = hdr[3];
  const bool has_info = hdr[4] != 0;
  std::vector<double> fr = read_vec<double>(f, (size_t)nf * 8);
  std::vector<int32_t> dof = read_vec<int32_t>(f, nf);
  std::vector<int32_t> f1 = read_vec<int32_t>(f, n1), s1 = read_vec<int32_t>(f, n1);
  std::vector<double> m1 = read_vec<double>(f, (size_t)n1 * 7), i1 = read_vec<double>(f, has_info ? (size_t)n1 * 36 : 0);
  std::vector<int32_t> f2 = read_vec<int32_t>(f, n2), s2 = read_vec<int32_t>(f, n2);
  std::vector<double> m2 = read_vec<double>(f, (size_t)n2 * 8), i2 = read_vec<double>(f, has_info ? (size_t)n2 * 49 : 0);
  std::vector<int32_t> fg = read_vec<int32_t>(f, ng);
  std::vector<double> mg = read_vec<double>(f, (size_t)ng * 7), ig = read_vec<double>(f, has_info ? (size_t)ng * 36 : 0);
  OptimizerPtr opt_ptr = Optimizer::create();
  if (!opt_ptr) { std::cerr << "Optimizer::create() returned null\n"; return 2; }
  opt_ptr->_config.maxIterations = hdr[5];
  BundleGraph g;
  g.cameraDOF = UPDATE_CAMERA_NONE;
  g.keyframes.resize(nf);
  for (int i = 0; i < nf; ++i) {
    const double* p = &fr[(size_t)i * 8];
    g.keyframes[i].estimation = SIM3(SO3(p[0], p[1], p[2], p[3]), Point3d(p[4], p[5], p[6]), p[7]);
    g.keyframes[i].dof = (KeyFrameEstimzationDOF)dof[i];
  }
  for (int k = 0; k < n1; ++k) {
    SE3Edge e;
    e.firstId = f1[k]; e.secondId = s1[k];
    const double* m = &m1[(size_t)k * 7];
    e.measurement = SE3(SO3(m[0], m[1], m[2], m[3]), Point3d(m[4], m[5], m[6]));
    e.information = has_info ? &i1[(size_t)k * 36] : NULL;
    g.se3Graph.push_back(e);
  }
  for (int k = 0; k < n2; ++k) {
    SIM3Edge e;
    e.firstId = f2[k]; e.secondId = s2[k];
    const double* m = &m2[(size_t)k * 8];
    e.measurement = SIM3(SO3(m[0], m[1], m[2], m[3]), Point3d(m[4], m[5], m[6]), m[7]);
    e.information = has_info ? &i2[(size_t)k * 49] : NULL;
    g.sim3Graph.push_back(e);
  }
  for (int k = 0; k < ng; ++k) {
    GPSEdge e;
    e.frameId = fg[k];
    const double* m = &mg[(size_t)k * 7];
    e.measurement = SE3(SO3(m[0], m[1], m[2], m[3]), Point3d(m[4], m[5], m[6]));
    e.information = has_info ? &ig[(size_t)k * 36] : NULL;
    g.gpsGraph.push_back(e);
  }
  // optional landmark section (the general BundleGraph): int32 {n_xyz, n_idp, n_obs_xyz, n_obs_idp, has_obs_info, sphere}, double
  // huber; xyz n x 3, free n (u8); idp: host n (i32), anchor n x 3, [idepth, sigma] n x 2, dof n (i32); then for each of the
  // two observation lists: point (i32), frame (i32), measurement n x 3, information n x 4 (if has_obs_info)
  int32_t lh[6] = {0, 0, 0, 0, 0, 0};
  std::vector<double> oinf_xyz, oinf_idp;
  if (f.read((char*)lh, sizeof(lh))) {
    double huber = 0;
    f.read((char*)&huber, 8);
    opt_ptr->_config.projectErrorHuberThreshold = huber;
    if (lh[5]) opt_ptr->_config.cameraProjectionType = PROJECTION_SPHERE;
    std::vector<double> xyz = read_vec<double>(f, (size_t)lh[0] * 3);
    std::vector<uint8_t> xfree = read_vec<uint8_t>(f, lh[0]);
    std::vector<int32_t> host = read_vec<int32_t>(f, lh[1]);
    std::vector<double> anchor = read_vec<double>(f, (size_t)lh[1] * 3), est = read_vec<double>(f, (size_t)lh[1] * 2);
    std::vector<int32_t> idof = read_vec<int32_t>(f, lh[1]);
    for (int i = 0; i < lh[0]; ++i)
      g.mappoints.push_back(std::make_pair(Point3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), xfree[i] != 0));
    for (int i = 0; i < lh[1]; ++i) {
      InvDepthEstimation v;
      v.frameId = host[i];
      v.anchor = Point3d(anchor[3 * i], anchor[3 * i + 1], anchor[3 * i + 2]);
      v.estimation = Point2d(est[2 * i], est[2 * i + 1]);
      v.dof = (InvDepthEstimationDOF)idof[i];
      g.invDepths.push_back(v);
    }
    for (int kind = 0; kind < 2; ++kind) {
      const int n = lh[2 + kind];
      std::vector<int32_t> pt = read_vec<int32_t>(f, n), fr2 = read_vec<int32_t>(f, n);
      std::vector<double> ms = read_vec<double>(f, (size_t)n * 3);
      std::vector<double>& inf = kind == 0 ? oinf_xyz : oinf_idp;
      inf = read_vec<double>(f, lh[4] ? (size_t)n * 4 : 0);
      for (int k = 0; k < n; ++k) {
        BundleEdge e;
        e.pointId = pt[k]; e.frameId = fr2[k];
        e.measurement = Point3d(ms[3 * k], ms[3 * k + 1], ms[3 * k + 2]);
        e.information = lh[4] ? &inf[(size_t)k * 4] : NULL;
        (kind == 0 ? g.mappointObserves : g.invDepthObserves).push_back(e);
      }
    }
  }
  const bool ok = opt_ptr->optimize(g);
  std::ofstream o(out, std::ios::binary);
  int32_t okv = ok ? 1 : 0;
  o.write((char*)&okv, 4);
  for (int i = 0; i < nf; ++i) {
    const SIM3& T = g.keyframes[i].estimation;
    SO3 r = T.get_rotation();
    Point3d t = T.get_translation();
    double p[8] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z, T.get_scale()};
    o.write((char*)p, sizeof(p));
  }
  for (size_t i = 0; i < g.mappoints.size(); ++i) {
    double p[3] = {g.mappoints[i].first.x, g.mappoints[i].first.y, g.mappoints[i].first.z};
    o.write((char*)p, sizeof(p));
  }
  for (size_t i = 0; i < g.invDepths.size(); ++i) {
    double p[2] = {g.invDepths[i].estimation.x, g.invDepths[i].estimation.y};
    o.write((char*)p, sizeof(p));
  }
  std::cout << "pose_graph_optimize=" << ok << std::endl;
  return ok ? 0 : 3;
}

// optimizeICP + fitSim3 + optimizePose in one go.  File: int32 {n, dof}; src n x 3; dst n x 3; then the tracking problem:
// int32 m; anchors1 m x 3; anchors2 m x 3; idepth m x 2; start pose 7 [qx qy qz qw tx ty tz].
static int run_align(const std::string& dir, const char* in, const char* out) {
  svar.GetString("OptimizerPlugin", "") = dir + "/libgslam_optimizer.so";
  std::ifstream f(in, std::ios::binary);
  int32_t hdr[2];
  f.read((char*)hdr, sizeof(hdr));
  const int n = hdr[0];
  std::vector<double> a = read_vec<double>(f, (size_t)n * 3), b = read_vec<double>(f, (size_t)n * 3);
  int32_t m;
  f.read((char*)&m, 4);
  std::vector<double> a1 = read_vec<double>(f, (size_t)m * 3), a2 = read_vec<double>(f, (size_t)m * 3), idp = read_vec<double>(f, (size_t)m * 2),
                      p0 = read_vec<double>(f, 7);
  OptimizerPtr opt_ptr = Optimizer::create();
  if (!opt_ptr) return 2;
  std::vector<std::pair<Point3d, Point3d> > pts(n);
  std::vector<std::pair<SE3, SE3> > traj(n);
  for (int k = 0; k < n; ++k) {
    pts[k] = std::make_pair(Point3d(a[3 * k], a[3 * k + 1], a[3 * k + 2]), Point3d(b[3 * k], b[3 * k + 1], b[3 * k + 2]));
    // rotations of the trajectory poses are irrelevant to fitSim3: give them something non-trivial
    traj[k] = std::make_pair(SE3(SO3::exp(Point3d(0.1 * k, 0.2, -0.1)), pts[k].first), SE3(SO3::exp(Point3d(0.3, -0.1 * k, 0.2)), pts[k].second));
  }
  SIM3 S1, S2;
  double info1[49], info2[49];
  const bool ok1 = opt_ptr->optimizeICP(pts, S1, (KeyFrameEstimzationDOF)hdr[1], info1);
  const bool ok2 = opt_ptr->fitSim3(traj, S2, (KeyFrameEstimzationDOF)hdr[1], info2);
  std::vector<std::pair<CameraAnchor, CameraAnchor> > mm(m);
  std::vector<IdepthEstimation> idv(m);
  for (int k = 0; k < m; ++k) {
    mm[k] = std::make_pair(Point3d(a1[3 * k], a1[3 * k + 1], a1[3 * k + 2]), Point3d(a2[3 * k], a2[3 * k + 1], a2[3 * k + 2]));
    idv[k] = IdepthEstimation(idp[2 * k], idp[2 * k + 1]);
  }
  SE3 rel(SO3(p0[0], p0[1], p0[2], p0[3]), Point3d(p0[4], p0[5], p0[6]));
  double info3[36];
  const bool ok3 = opt_ptr->optimizePose(mm, idv, rel, UPDATE_KF_SE3, info3);
  std::ofstream o(out, std::ios::binary);
  int32_t oks[3] = {ok1 ? 1 : 0, ok2 ? 1 : 0, ok3 ? 1 : 0};
  o.write((char*)oks, sizeof(oks));
  auto w8 = [&o](const SIM3& T) {
    SO3 r = T.get_rotation();
    Point3d t = T.get_translation();
    double p[8] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z, T.get_scale()};
    o.write((char*)p, sizeof(p));
  };
  w8(S1);
  o.write((char*)info1, sizeof(info1));
  w8(S2);
  o.write((char*)info2, sizeof(info2));
  w8(SIM3(rel, 1.0));
  o.write((char*)info3, sizeof(info3));
  std::cout << "optimizeICP=" << ok1 << " fitSim3=" << ok2 << " optimizePose=" << ok3 << std::endl;
  return ok1 && ok2 && ok3 ? 0 : 3;
}

// sphere != 0: the measurements are 3-vectors (bearings, any length) and the optimiser runs under PROJECTION_SPHERE
static int run_pnp(const std::string& dir, const char* in, const char* out, int sphere = 0) {
  svar.GetString("OptimizerPlugin", "") = dir + "/libgslam_optimizer.so";
  std::ifstream f(in, std::ios::binary);
  int32_t n;
  f.read((char*)&n, 4);
  const int md = sphere ? 3 : 2;
  std::vector<double> X = read_vec<double>(f, (size_t)n * 3), m = read_vec<double>(f, (size_t)n * md),
                      p = read_vec<double>(f, 7);
  OptimizerPtr opt_ptr = Optimizer::create();
  if (!opt_ptr) return 2;
  if (sphere) opt_ptr->_config.cameraProjectionType = PROJECTION_SPHERE;
  std::vector<std::pair<Point3d, CameraAnchor> > matches(n);
  for (int k = 0; k < n; ++k)
    matches[k] = std::make_pair(Point3d(X[3 * k], X[3 * k + 1], X[3 * k + 2]),
                                sphere ? Point3d(m[3 * k], m[3 * k + 1], m[3 * k + 2]) : Point3d(m[2 * k], m[2 * k + 1], 1.0));
  SE3 pose(SO3(p[0], p[1], p[2], p[3]), Point3d(p[4], p[5], p[6]));
  double info[36];
  const bool ok = opt_ptr->optimizePnP(matches, pose, UPDATE_KF_SE3, info);
  std::ofstream o(out, std::ios::binary);
  int32_t okv = ok ? 1 : 0;
  o.write((char*)&okv, 4);
  SO3 r = pose.get_rotation();
  Point3d t = pose.get_translation();
  double q[7] = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
  o.write((char*)q, sizeof(q));
  o.write((char*)info, sizeof(info));
  return ok ? 0 : 3;
}

static int run_orb(const std::string& dir, int w, int h, int ch, const char* in, const char* out, int K) {
  svar.GetString("FeatureDetectorPlugin", "") = dir + "/libgslam_featuredetector.so";
  std::vector<uchar> img((size_t)w * h * ch);
  std::ifstream f(in, std::ios::binary);
  f.read((char*)img.data(), img.size());
  FeatureDetectorPtr det = FeatureDetector::create();
  if (!det) { std::cerr << "FeatureDetector::create() returned null\n"; return 2; }
  det->_config.nFeatures = K;
  GImage image(h, w, ch == 1 ? GImageType<uchar, 1>::Type : (ch == 3 ? GImageType<uchar, 3>::Type : GImageType<uchar, 4>::Type),
               img.data(), false);
  std::vector<KeyPoint> kps;
  GImage desc;
  const bool ok = det->detectAndCompute(image, kps, desc);
  std::vector<std::pair<int, int> > matches;
  std::vector<uchar> mask;
  det->_config.matchCrossCheck = true;
  const bool okm = ok && det->match(desc, desc, matches, &mask);
  std::ofstream o(out, std::ios::binary);
  int32_t hdr[4] = {ok ? 1 : 0, (int32_t)kps.size(), okm ? 1 : 0, (int32_t)matches.size()};
  o.write((char*)hdr, sizeof(hdr));
  if (!kps.empty()) {
    o.write((char*)kps.data(), kps.size() * sizeof(KeyPoint));
    o.write((char*)desc.data, (size_t)desc.rows * 32);
    o.write((char*)matches.data(), matches.size() * sizeof(std::pair<int, int>));
  }
  std::cout << "detectAndCompute=" << ok << " keypoints=" << kps.size() << " match=" << okm << " matches="
            << matches.size() << " desc=" << desc.rows << "x" << desc.cols << std::endl;
  return ok && okm ? 0 : 3;
}

static int gimage_type(int ch) {
  return ch == 1 ? GImageType<uchar, 1>::Type : (ch == 3 ? GImageType<uchar, 3>::Type : GImageType<uchar, 4>::Type);
}

static void write_frame_records(std::ofstream& o, const std::vector<std::vector<KeyPoint> >& kps, const std::vector<GImage>& desc) {
  for (size_t i = 0; i < kps.size(); ++i) {
    int32_t n = (int32_t)kps[i].size();
    o.write((char*)&n, 4);
    if (n) {
      o.write((char*)kps[i].data(), (size_t)n * sizeof(KeyPoint));
      o.write((char*)desc[i].data, (size_t)n * 32);
    }
  }
}

// Batch and asynchronous entries of the FeatureDetector interface: three result sets (per-frame, batch, async) for the
// same frames; the test compares each with the oracle.
static int run_orb_batch(const std::string& dir, int w, int h, int ch, int n, const char* in, const char* out, int K) {
  svar.GetString("FeatureDetectorPlugin", "") = dir + "/libgslam_featuredetector.so";
  svar.GetInt("FeatureDetectorHIP.BatchChunk", 16) = 5;  // several chunks + a partial one for small n
  std::vector<uchar> raw((size_t)w * h * ch * n);
  std::ifstream f(in, std::ios::binary);
  f.read((char*)raw.data(), raw.size());
  FeatureDetectorPtr det = FeatureDetector::create();
  if (!det) { std::cerr << "FeatureDetector::create() returned null\n"; return 2; }
  det->_config.nFeatures = K;
  std::vector<GImage> images;
  for (int i = 0; i < n; ++i) images.push_back(GImage(h, w, gimage_type(ch), raw.data() + (size_t)i * w * h * ch, false));
  std::vector<std::vector<KeyPoint> > k1(n), k2, k3(n);
  std::vector<GImage> d1(n), d2, d3(n);
  bool ok = true;
  for (int i = 0; i < n; ++i) ok = ok && det->detectAndCompute(images[i], k1[i], d1[i]);
  const bool okb = det->detectAndComputeBatch(images, k2, d2);
  // async: keep asyncDepth() tickets in flight, collect the oldest
  bool oka = det->asyncDepth() > 0;
  std::vector<long> tickets;
  size_t done = 0;
  for (int i = 0; i < n && oka; ++i) {
    if ((int)(tickets.size() - done) == det->asyncDepth()) {
      oka = det->collect(tickets[done], k3[done], d3[done]);
      ++done;
    }
    const long t = det->submit(images[i]);
    oka = oka && t >= 0;
    tickets.push_back(t);
  }
  for (; done < tickets.size() && oka; ++done) oka = det->collect(tickets[done], k3[done], d3[done]);
  std::ofstream o(out, std::ios::binary);
  int32_t hdr[4] = {ok ? 1 : 0, okb ? 1 : 0, oka ? 1 : 0, n};
  o.write((char*)hdr, sizeof(hdr));
  write_frame_records(o, k1, d1);
  if (okb) write_frame_records(o, k2, d2);
  if (oka) write_frame_records(o, k3, d3);
  // matchBatch: every ordered pair of the frames in ONE call (with and without cross-checking) against match() pair by pair
  bool okm = ok;
  size_t n_matches = 0;
  for (int cc = 0; cc < 2 && okm; ++cc) {
    det->_config.matchCrossCheck = cc != 0;
    std::vector<std::pair<int, int> > pairs;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        if (i != j) pairs.push_back(std::make_pair(i, j));
    std::vector<std::vector<std::pair<int, int> > > mb;
    okm = det->matchBatch(d1, pairs, mb) && mb.size() == pairs.size();
    for (size_t p = 0; p < pairs.size() && okm; ++p) {
      std::vector<std::pair<int, int> > m1;
      okm = det->match(d1[pairs[p].first], d1[pairs[p].second], m1) && m1 == mb[p];
      n_matches += m1.size();
    }
  }
  det->_config.matchCrossCheck = false;
  std::cout << "orbbatch single=" << ok << " batch=" << okb << " async=" << oka << " depth=" << det->asyncDepth()
            << " matchBatch=" << okm << " (" << n_matches << " matches)" << std::endl;
  return ok && okb && oka && okm ? 0 : 3;
}

static void pct(std::vector<double>& v, double* p50, double* p99) {
  std::sort(v.begin(), v.end());
  *p50 = v[v.size() / 2];
  *p99 = v[std::min(v.size() - 1, (size_t)(v.size() * 0.99))];
}

// What one tracking step costs a GSLAM application that calls the plugins frame by frame: detectAndCompute on the new
// frame, match against the previous one, optimizePnP on 3D-2D correspondences (GSLAM/plugins/play/main.cpp:99-155 feeds
// frames one at a time; evaluation/metric_time/main.cpp:4-34 measures exactly this per-frame time).
static int run_lat(const std::string& dir, int w, int h, int n, const char* in, int K, int iters) {
  svar.GetString("FeatureDetectorPlugin", "") = dir + "/libgslam_featuredetector.so";
  svar.GetString("OptimizerPlugin", "") = dir + "/libgslam_optimizer.so";
  std::vector<uchar> raw((size_t)w * h * n);
  std::ifstream f(in, std::ios::binary);
  f.read((char*)raw.data(), raw.size());
  FeatureDetectorPtr det = FeatureDetector::create();
  OptimizerPtr opt = Optimizer::create();
  if (!det || !opt) { std::cerr << "plugin create() returned null\n"; return 2; }
  det->_config.nFeatures = K;
  // a PnP problem of the size tracking sees: 300 points in front of the camera, exact projections, perturbed start
  std::vector<std::pair<Point3d, CameraAnchor> > m3d;
  unsigned long long rs = 88172645463325252ull;
  auto rnd = [&rs]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; };
  const SE3 truth(SO3::exp(Point3d(0.02, -0.03, 0.01)), Point3d(0.1, -0.05, 0.2));
  for (int k = 0; k < 300; ++k) {
    const Point3d X(4 * rnd() - 2, 3 * rnd() - 1.5, 4 + 4 * rnd());
    const Point3d c = truth.inverse() * X;
    m3d.push_back(std::make_pair(X, Point3d(c.x / c.z, c.y / c.z, 1.0)));
  }
  std::vector<double> t_det, t_match, t_pnp, t_all;
  std::vector<KeyPoint> kps, prev_kps;
  GImage desc, prev_desc;
  typedef std::chrono::steady_clock clk;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  bool ok = true;
  for (int it = 0; it < iters + 20 && ok; ++it) {
    GImage img(h, w, GImageType<uchar, 1>::Type, raw.data() + (size_t)(it % n) * w * h, false);
    const clk::time_point t0 = clk::now();
    ok = det->detectAndCompute(img, kps, desc);
    const clk::time_point t1 = clk::now();
    std::vector<std::pair<int, int> > matches;
    if (ok && !prev_desc.empty()) ok = det->match(desc, prev_desc, matches);
    const clk::time_point t2 = clk::now();
    SE3 pose;
    ok = ok && opt->optimizePnP(m3d, pose);
    const clk::time_point t3 = clk::now();
    if (it >= 20) {
      t_det.push_back(ms(t0, t1));
      t_match.push_back(ms(t1, t2));
      t_pnp.push_back(ms(t2, t3));
      t_all.push_back(ms(t0, t3));
    }
    prev_desc = desc;
  }
  if (!ok) { std::cerr << "a plugin call failed\n"; return 3; }
  // asynchronous throughput: asyncDepth() frames in flight, frame by frame
  double async_fps = 0;
  if (det->asyncDepth() > 0) {
    std::vector<long> tickets;
    size_t done = 0;
    const int total = iters;
    const clk::time_point a0 = clk::now();
    for (int it = 0; it < total; ++it) {
      GImage img(h, w, GImageType<uchar, 1>::Type, raw.data() + (size_t)(it % n) * w * h, false);
      if ((int)(tickets.size() - done) == det->asyncDepth()) det->collect(tickets[done++], kps, desc);
      tickets.push_back(det->submit(img));
    }
    for (; done < tickets.size(); ++done) det->collect(tickets[done], kps, desc);
    async_fps = total / (ms(a0, clk::now()) * 1e-3);
  }
  double a, b;
  std::cout << "lat w=" << w << " h=" << h << " K=" << K << " iters=" << iters << " keypoints=" << kps.size();
  pct(t_det, &a, &b);   std::cout << " detect_p50_ms=" << a << " detect_p99_ms=" << b;
  pct(t_match, &a, &b); std::cout << " match_p50_ms=" << a << " match_p99_ms=" << b;
  pct(t_pnp, &a, &b);   std::cout << " pnp_p50_ms=" << a << " pnp_p99_ms=" << b;
  pct(t_all, &a, &b);   std::cout << " step_p50_ms=" << a << " step_p99_ms=" << b;
  std::cout << " async_frames_per_s=" << async_fps << std::endl;
  return 0;
}

// Vocabulary: the plugin's subclass against the reference's own base-class transform, in the same process.
static int run_bow(const std::string& dir, const char* gbow, const char* descf, int n, int levelsup, const char* out) {
  typedef std::shared_ptr<Vocabulary> (*factory_t)(const char*);
  std::shared_ptr<SharedLibrary> lib = Registry::get(dir + "/libgslam_vocabulary.so");
  if (!lib) { std::cerr << "cannot load libgslam_vocabulary.so\n"; return 2; }
  factory_t f = (factory_t)lib->getSymbol("createVocabularyInstance");
  if (!f) return 2;
  std::shared_ptr<Vocabulary> gpu = f(gbow);
  Vocabulary cpu;
  if (!gpu || !cpu.load(std::string(gbow))) { std::cerr << "vocabulary load failed\n"; return 2; }
  const int W = (int)(cpu.m_nodeDescriptors.cols * cpu.m_nodeDescriptors.elemSize());  // descriptor bytes of this vocabulary
  const bool is_f32 = cpu.m_nodeDescriptors.type() == GImageType<float>::Type;
  std::vector<uchar> d((size_t)n * W);
  std::ifstream fi(descf, std::ios::binary);
  fi.read((char*)d.data(), d.size());
  TinyMat features(n, is_f32 ? W / 4 : W, is_f32 ? (int)GImageType<float>::Type : (int)GImageType<uchar>::Type, d.data(), false);
  BowVector bg, bc;
  FeatureVector fg, fc;
  gpu->transform(features, bg, fg, levelsup);
  cpu.transform(features, bc, fc, levelsup);
  BowVector bg2, bc2;
  gpu->transform(features, bg2);
  cpu.transform(features, bc2);
  bool same = bg == bc && fg == fc && bg2 == bc2;
  // the std::vector<TinyMat> overload (one feature per element) and the single-feature transform -> WordId, against the
  // base class on the same features (Vocabulary.h:183,200)
  {
    std::vector<TinyMat> list;
    for (int i = 0; i < n; ++i) list.push_back(TinyMat(1, is_f32 ? W / 4 : W, is_f32 ? (int)GImageType<float>::Type : (int)GImageType<uchar>::Type, d.data() + (size_t)i * W, false));
    BowVector bgl, bcl;
    FeatureVector fgl, fcl;
    gpu->transform(list, bgl, fgl, levelsup);
    cpu.transform(list, bcl, fcl, levelsup);
    int word_bad = 0;
    for (int i = 0; i < std::min(n, 200); ++i)
      if (gpu->transform(list[i]) != cpu.transform(list[i])) ++word_bad;
    typedef int (*fail_t)(const Vocabulary*);
    fail_t fc_ = (fail_t)lib->getSymbol("vocabularyFailureCount");
    const int fails = fc_ ? fc_(gpu.get()) : -2;
    std::cout << "list overload equal=" << (bgl == bcl && fgl == fcl && bgl == bg) << " single-feature word mismatches=" << word_bad
              << " failures=" << fails << std::endl;
    same = same && bgl == bcl && fgl == fcl && word_bad == 0 && fails == 0;
  }
  // batched scoring (scoreVocabularyBatch) against the reference's own score() on a database of sub-images: windows
  // of the same descriptor list, transformed by the reference itself
  typedef bool (*score_t)(const Vocabulary*, const BowVector*, const BowVector* const*, int, double*);
  score_t sb = (score_t)lib->getSymbol("scoreVocabularyBatch");
  int score_bad = -1;
  if (sb) {
    std::vector<BowVector> db;
    for (int w0 = 0; w0 + 50 <= n && db.size() < 40; w0 += n / 40 + 1) {
      const int wn = std::min(n - w0, 50 + (int)db.size() * 17);
      TinyMat sub(wn, is_f32 ? W / 4 : W, is_f32 ? (int)GImageType<float>::Type : (int)GImageType<uchar>::Type, d.data() + (size_t)w0 * W, false);
      BowVector v;
      cpu.transform(sub, v);
      db.push_back(v);
    }
    db.push_back(BowVector());  // an empty vector
    std::vector<const BowVector*> ptr;
    for (auto& v : db) ptr.push_back(&v);
    std::vector<double> sc(db.size(), -1.0);
    score_bad = sb(gpu.get(), &bc2, ptr.data(), (int)ptr.size(), sc.data()) ? 0 : (int)db.size();
    for (size_t j = 0; j < db.size() && score_bad >= 0; ++j) {
      const double e = cpu.m_scoring_object->score(bc2, db[j]);  // Vocabulary::score itself is declared but never defined
      const bool kl = cpu.getScoringType() == Vocabulary::KL;
      if (kl ? std::fabs(sc[j] - e) > 1e-6 * std::max(1.0, std::fabs(e)) : sc[j] != e) ++score_bad;
    }
    std::cout << "score batch: " << db.size() << " database vectors, mismatches=" << score_bad << std::endl;
  }
  same = same && score_bad == 0;
  std::ofstream o(out, std::ios::binary);
  int32_t hdr[4] = {same ? 1 : 0, (int32_t)bg.size(), (int32_t)fg.size(), (int32_t)bc.size()};
  o.write((char*)hdr, sizeof(hdr));
  for (auto& kv : bg) {
    uint64_t id = kv.first;
    float val = kv.second;
    o.write((char*)&id, 8);
    o.write((char*)&val, 4);
  }
  std::cout << "bow gpu==reference:" << same << " words=" << bg.size() << " nodes=" << fg.size()
            << " score(self)=" << gpu->m_scoring_object->score(bg, bc) << std::endl;
  return same ? 0 : 3;
}

// UndistorterHIP against GSLAM::Undistorter in the same process, on the pixels the reference defines.
static int run_undist(int ch, const char* imgf) {
  std::vector<double> pin = {320, 240, 260, 262, 161.5, 118.2, -0.31, 0.11, 0.001, -0.0007, -0.02};
  std::vector<double> pout = {320, 240, 200, 200, 160, 120};
  Camera cin(pin), cout_(pout);
  UndistorterHIP gpu(cin, cout_);
  std::streambuf* old = std::cout.rdbuf(nullptr);
  Undistorter cpu(cin, cout_);
  std::cout.rdbuf(old);
  if (!gpu.valid() || !cpu.valid()) { std::cerr << "undistorter invalid\n"; return 2; }
  std::vector<uchar> img((size_t)320 * 240 * ch);
  std::ifstream f(imgf, std::ios::binary);
  f.read((char*)img.data(), img.size());
  GImage in(240, 320, ch == 1 ? GImageType<uchar, 1>::Type : GImageType<uchar, 3>::Type, img.data(), false);
  UndistorterImpl tab(cin, cout_);  // tables, to know which pixels the reference defines
  long bad = 0, checked = 0;
  for (int fast = 0; fast < 2; ++fast) {
    GImage a, b;
    const bool oka = fast ? gpu.undistortFast(in, a) : gpu.undistort(in, a);
    const bool okb = fast ? cpu.undistortFast(in, b) : cpu.undistort(in, b);
    if (!oka || !okb) return 3;
    const size_t n_in = (size_t)320 * 240;
    for (int i = 0; i < 320 * 240; ++i) {
      bool defined;
      if (fast) defined = ch == 1 ? tab.remapFast[i] > 0 : tab.remapFast[i] >= 0;
      else {
        defined = ch == 1 ? true : tab.remapX[i] > 0;
        for (int t = 0; t < 4; ++t) defined = defined && (size_t)tab.remapIdx[4 * i + t] < n_in;
      }
      if (!defined) continue;
      ++checked;
      for (int j = 0; j < ch; ++j) bad += a.data[(size_t)i * ch + j] != b.data[(size_t)i * ch + j];
    }
  }
  std::cout << "undistort gpu==reference mismatches=" << bad << " checked=" << checked << std::endl;
  return bad == 0 && checked > 100000 ? 0 : 3;
}

// A minimal frame type that stores what MapFrame::setKeyPoints hands over (the gmap plugin's MapFrame does the same,
// plugins/gmap/MapFrame.cpp:211-247).  Published as std::shared_ptr<MapFrame> (Messenger payloads are exact-typed).
class HostFrame : public MapFrame {
 public:
  HostFrame(FrameID id, double t, const GImage& img) : MapFrame(id, t), img_(img) {}
  std::string type() const override { return "HostFrame"; }
  int cameraNum() const override { return 1; }
  int imageChannels(int) const override { return IMAGE_GRAY; }
  GImage getImage(int, int) override { return img_; }
  int keyPointNum() const override { return (int)kps_.size(); }
  bool setKeyPoints(const std::vector<KeyPoint>& k, const GImage& d) override {
    kps_ = k;
    desc_ = d.clone();
    return true;
  }
  bool getKeyPoints(std::vector<KeyPoint>& k) const override {
    k = kps_;
    return true;
  }
  GImage getDescriptor(int idx) const override { return idx < 0 ? desc_ : desc_.row(idx); }

 private:
  GImage img_, desc_;
  std::vector<KeyPoint> kps_;
};

static int run_app(const std::string& dir, int w, int h, int n, const char* framesf, const char* out, int K) {
  svar.GetString("FeatureDetectorPlugin", "") = dir + "/libgslam_featuredetector.so";
  svar.GetInt("orbhip.nFeatures", 1000) = K;
  // --- what GSLAM/gslam/main.cpp does for every application name on its command line
  Svar app = Registry::load(dir + "/libgslam_orbhip.so");
  if (app.isUndefined()) { std::cerr << "cannot load libgslam_orbhip.so\n"; return 2; }
  Svar run = app["gslam"]["apps"]["orbhip"];
  if (!run.isFunction()) { std::cerr << "plugin has no run function\n"; return 2; }
  Svar fsink = app["gslam"]["setGlobalLogSinks"], fmsg = app["gslam"]["setGlobalMessenger"];
  if (fsink.isFunction()) fsink(getLogSinksGlobal());
  if (fmsg.isFunction()) fmsg(messenger);
  std::thread th([run]() { run(svar); });

  // --- collect what the application publishes
  std::mutex mu;
  std::vector<FramePtr> got;
  std::vector<int> nmatch;
  Subscriber s1 = messenger.subscribe("orbhip/curframe", 0, [&](FramePtr fr) {
    std::lock_guard<std::mutex> l(mu);
    got.push_back(fr);
  });
  Subscriber s2 = messenger.subscribe("orbhip/matches", 0, [&](Svar m) {
    std::lock_guard<std::mutex> l(mu);
    nmatch.push_back(m["matches"].castAs<int>());
  });
  // wait until the application has subscribed, then play the frames like plugins/play does
  Publisher pub = messenger.advertise<MapFrame>("dataset/frame", 0);
  for (int i = 0; i < 200 && pub.getNumSubscribers() == 0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
  if (pub.getNumSubscribers() == 0) { std::cerr << "orbhip never subscribed\n"; return 3; }
  std::vector<uchar> raw((size_t)w * h * n);
  std::ifstream f(framesf, std::ios::binary);
  f.read((char*)raw.data(), raw.size());
  for (int i = 0; i < n; ++i) {
    GImage img(h, w, GImageType<uchar, 1>::Type, raw.data() + (size_t)i * w * h, true);
    pub.publish(FramePtr(new HostFrame(i + 1, 0.05 * i, img)));
  }
  messenger.publish("messenger/stop", true);
  th.join();
  std::ofstream o(out, std::ios::binary);
  int32_t hdr[2] = {(int32_t)got.size(), (int32_t)nmatch.size()};
  o.write((char*)hdr, sizeof(hdr));
  for (size_t i = 0; i < got.size(); ++i) {
    std::vector<KeyPoint> kps = got[i]->getKeyPoints();
    GImage d = got[i]->getDescriptor(-1);
    int32_t rec[3] = {(int32_t)got[i]->id(), (int32_t)kps.size(), i < nmatch.size() ? nmatch[i] : -1};
    o.write((char*)rec, sizeof(rec));
    if (!kps.empty()) {
      o.write((char*)kps.data(), kps.size() * sizeof(KeyPoint));
      o.write((char*)d.data, (size_t)d.rows * 32);
    }
  }
  std::cout << "app frames_out=" << got.size() << " match_msgs=" << nmatch.size() << std::endl;
  return (int)got.size() == n ? 0 : 3;
}

static int run_est(const std::string& dir, int model, int n, const char* ptsf, double thr, const char* out) {
  svar.GetString("EstimatorPlugin", "") = dir + "/libgslam_estimator.so";
  EstimatorPtr est = Estimator::create();
  if (!est) { std::cerr << "Estimator::create() returned null\n"; return 2; }
  std::vector<double> raw((size_t)n * 4);
  std::ifstream f(ptsf, std::ios::binary);
  f.read((char*)raw.data(), raw.size() * 8);
  std::vector<Point2d> a(n), b(n);
  for (int i = 0; i < n; ++i) {
    a[i] = Point2d(raw[4 * i], raw[4 * i + 1]);
    b[i] = Point2d(raw[4 * i + 2], raw[4 * i + 3]);
  }
  std::vector<uchar> mask;
  double m[9] = {0};
  bool ok = false;
  if (model == 0) {
    Homography2D H;
    ok = est->findHomography(&H, a, b, H4_Point | RANSAC, thr, 0.99, &mask);
    for (int i = 0; i < 9; ++i) m[i] = H.data()[i];
  } else if (model == 1) {
    Affine2D A;
    ok = est->findAffine2D(&A, a, b, A3_Point | RANSAC, thr, 0.99, &mask);
    for (int i = 0; i < 6; ++i) m[i] = A.data()[i];
  } else if (model == 4) {
    Essential E;
    ok = est->findEssentialMatrix(&E, a, b, E5_Nister | RANSAC, thr, 0.99, &mask);
    for (int i = 0; i < 9; ++i) m[i] = E.data()[i];
  } else {
    Fundamental F;
    ok = est->findFundamental(&F, a, b, F8_Point | RANSAC, thr, 0.99, &mask);
    for (int i = 0; i < 9; ++i) m[i] = F.data()[i];
  }
  SE3 pose;
  std::vector<Point3d> p3;
  const bool unsupported = est->findPnP(&pose, p3, a);  // mismatched sizes: the one call that must refuse
  // the sampling flags of EstimatorMethod (Estimator.h:86-89): NOSAMPLE (2 << 8, distinguishable next to a model id) and LMEDS
  // (1 << 8 == F8_Point: passed bare; not meaningful for findFundamental, which reads it as F8_Point + RANSAC)
  auto call = [&](int method, double* mm, std::vector<uchar>* mk) {
    for (int i = 0; i < 9; ++i) mm[i] = 0;
    if (model == 0) { Homography2D H; const bool r = est->findHomography(&H, a, b, method, thr, 0.99, mk); for (int i = 0; i < 9; ++i) mm[i] = H.data()[i]; return r; }
    if (model == 1) { Affine2D A; const bool r = est->findAffine2D(&A, a, b, method, thr, 0.99, mk); for (int i = 0; i < 6; ++i) mm[i] = A.data()[i]; return r; }
    if (model == 4) { Essential E; const bool r = est->findEssentialMatrix(&E, a, b, method, thr, 0.99, mk); for (int i = 0; i < 9; ++i) mm[i] = E.data()[i]; return r; }
    Fundamental F; const bool r = est->findFundamental(&F, a, b, method, thr, 0.99, mk); for (int i = 0; i < 9; ++i) mm[i] = F.data()[i]; return r;
  };
  double m_nos[9], m_lmeds[9];
  std::vector<uchar> mask_nos, mask_lmeds;
  const int model_id[5] = {H4_Point, A3_Point, F8_Point, 0, E5_Nister};
  const bool ok_nos = call(model_id[model] | NOSAMPLE, m_nos, &mask_nos);
  const bool ok_lmeds = model == 2 ? false : call(LMEDS, m_lmeds, &mask_lmeds);
  std::ofstream o(out, std::ios::binary);
  int32_t hdr[2] = {ok ? 1 : 0, (int32_t)mask.size()};
  o.write((char*)hdr, sizeof(hdr));
  o.write((char*)m, sizeof(m));
  if (!mask.empty()) o.write((char*)mask.data(), mask.size());
  for (int v = 0; v < 2; ++v) {
    const int32_t h2[2] = {(v == 0 ? ok_nos : ok_lmeds) ? 1 : 0, (int32_t)(v == 0 ? mask_nos : mask_lmeds).size()};
    o.write((char*)h2, sizeof(h2));
    o.write((char*)(v == 0 ? m_nos : m_lmeds), 72);
    const std::vector<uchar>& mk = v == 0 ? mask_nos : mask_lmeds;
    if (!mk.empty()) o.write((char*)mk.data(), mk.size());
  }
  std::cout << "estimator " << est->type() << " ok=" << ok << " unsupported_paths=" << unsupported << std::endl;
  return ok && !unsupported ? 0 : 3;
}

// findSIM3 (5) / findPlane (6) / findPnP (7) / trianglate (8) through Estimator::create(); input n x 6 doubles per row:
// SIM3 (from xyz, to xyz), plane (xyz, unused), PnP (object xyz, normalised uv, unused), trianglate (ref dir, cur dir)
// preceded, for mode 8, by the ref -> cur pose as 7 doubles.
static int run_est3(const std::string& dir, int model, int n, const char* ptsf, double thr, const char* out) {
  svar.GetString("EstimatorPlugin", "") = dir + "/libgslam_estimator.so";
  EstimatorPtr est = Estimator::create();
  if (!est) { std::cerr << "Estimator::create() returned null\n"; return 2; }
  std::ifstream f(ptsf, std::ios::binary);
  double pose7[7] = {0, 0, 0, 1, 0, 0, 0};
  if (model == 8) f.read((char*)pose7, sizeof(pose7));
  std::vector<double> raw((size_t)n * 6);
  f.read((char*)raw.data(), raw.size() * 8);
  std::vector<Point3d> a(n), b(n);
  std::vector<Point2d> uv(n);
  for (int i = 0; i < n; ++i) {
    a[i] = Point3d(raw[6 * i], raw[6 * i + 1], raw[6 * i + 2]);
    b[i] = Point3d(raw[6 * i + 3], raw[6 * i + 4], raw[6 * i + 5]);
    uv[i] = Point2d(raw[6 * i + 3], raw[6 * i + 4]);
  }
  std::vector<uchar> mask;
  std::vector<double> m;
  bool ok = false;
  if (model == 5) {
    SIM3 S;
    ok = est->findSIM3(&S, a, b, S3_Horn | RANSAC, thr, 0.99, &mask);
    const SO3 r = S.get_rotation();
    const Point3d t = S.get_translation();
    m = {r.x, r.y, r.z, r.w, t.x, t.y, t.z, S.get_scale()};
  } else if (model == 6) {
    SE3 P;
    ok = est->findPlane(&P, a, P3_Plane | RANSAC, thr, 0.99, &mask);
    const SO3 r = P.get_rotation();
    const Point3d t = P.get_translation(), z = r * Point3d(0, 0, 1);
    m = {z.x, z.y, z.z, t.x, t.y, t.z};  // normal, point on the plane
  } else if (model == 7) {
    SE3 T;
    ok = est->findPnP(&T, a, uv, P3_ITERATIVE | RANSAC, thr, 0.99, &mask);
    const SO3 r = T.get_rotation();
    const Point3d t = T.get_translation();
    m = {r.x, r.y, r.z, r.w, t.x, t.y, t.z};
  } else {
    const SE3 T(SO3(pose7[0], pose7[1], pose7[2], pose7[3]), Point3d(pose7[4], pose7[5], pose7[6]));
    ok = true;
    for (int i = 0; i < n; ++i) {
      Point3d X(0, 0, 0);
      const bool one = est->trianglate(&X, T, a[i], b[i]);
      mask.push_back(one ? 1 : 0);
      m.push_back(X.x); m.push_back(X.y); m.push_back(X.z);
    }
  }
  std::ofstream o(out, std::ios::binary);
  int32_t hdr[3] = {ok ? 1 : 0, (int32_t)mask.size(), (int32_t)m.size()};
  o.write((char*)hdr, sizeof(hdr));
  o.write((char*)m.data(), m.size() * 8);
  if (!mask.empty()) o.write((char*)mask.data(), mask.size());
  std::cout << "estimator3 " << est->type() << " model=" << model << " ok=" << ok << std::endl;
  return ok ? 0 : 3;
}

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  const std::string mode = argv[1], dir = argv[2];
  // integer svar options for the plugins under test: GSLAM_HOST_SVAR="FeatureDetectorHIP.Steering=1;FeatureDetectorHIP.Distribution=1"
  if (const char* e = getenv("GSLAM_HOST_SVAR")) {
    std::stringstream ss(e);
    std::string kv;
    while (std::getline(ss, kv, ';')) {
      const size_t eq = kv.find('=');
      if (eq != std::string::npos) svar.GetInt(kv.substr(0, eq), 0) = atoi(kv.c_str() + eq + 1);
    }
  }
  if (mode == "ba" && argc >= 5)
    return run_ba(dir, argv[3], argv[4], argc >= 6 ? atof(argv[5]) : 1.0, argc >= 7 ? atoi(argv[6]) : -1);
  if (mode == "magin" && argc >= 5) return run_magin(dir, argv[3], argv[4]);
  if (mode == "calib" && argc >= 5) return run_calib(dir, argv[3], argv[4]);
  if (mode == "pnp" && argc >= 5) return run_pnp(dir, argv[3], argv[4], argc >= 6 ? atoi(argv[5]) : 0);
  if (mode == "pg" && argc >= 5) return run_pg(dir, argv[3], argv[4]);
  if (mode == "align" && argc >= 5) return run_align(dir, argv[3], argv[4]);
  if (mode == "est" && argc >= 8) return run_est(dir, atoi(argv[3]), atoi(argv[4]), argv[5], atof(argv[6]), argv[7]);
  if (mode == "est3" && argc >= 8) return run_est3(dir, atoi(argv[3]), atoi(argv[4]), argv[5], atof(argv[6]), argv[7]);
  if (mode == "app" && argc >= 9)
    return run_app(dir, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argv[6], argv[7], atoi(argv[8]));
  if (mode == "undist" && argc >= 5) return run_undist(atoi(argv[3]), argv[4]);
  if (mode == "bow" && argc >= 8) return run_bow(dir, argv[3], argv[4], atoi(argv[5]), atoi(argv[6]), argv[7]);
  if (mode == "orbbatch" && argc >= 10)
    return run_orb_batch(dir, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), argv[7], argv[8], atoi(argv[9]));
  if (mode == "lat" && argc >= 9) return run_lat(dir, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argv[6], atoi(argv[7]), atoi(argv[8]));
  if (mode == "orb" && argc >= 9)
    return run_orb(dir, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argv[6], argv[7], atoi(argv[8]));
  return 1;
}
