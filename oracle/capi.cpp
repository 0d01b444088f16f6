// ORACLE (test infrastructure only — never linked into or called by the product path).
// Flat C API over the CPU restatement so that tests/ and bench.py's cpu_baseline leg can drive it via ctypes.
#include <array>
#include "oracle.h"

using namespace ovio;
using namespace om;

namespace ovio { void marg_finish(Estimator &e, Mat &A, std::vector<double> &b, int m, int n); }  // backend.cpp

extern "C" {

int ovio_config_size() { return (int)sizeof(Config); }
void ovio_config_default(Config *c) { *c = Config(); }

// ---------------------------------------------------------------- full pipeline (tracker + estimator + nodelet glue)
void *ovio_pipeline_create(const Config *c) { return new Pipeline(*c); }
void ovio_pipeline_destroy(void *h) { delete (Pipeline *)h; }
void ovio_pipeline_restart(void *h) { ((Pipeline *)h)->restart(); }
// Estimator::setReloFrame (estimator.cpp:1728-1747): match_points[n][3] = (x, y, feature id), relo_r row-major
void ovio_set_relo_frame(void *h, double stamp, int index, int n, const double *mp, const double *relo_t, const double *relo_r) {
    std::vector<std::array<double, 3>> v(n);
    for (int i = 0; i < n; i++) v[i] = {mp[3 * i], mp[3 * i + 1], mp[3 * i + 2]};
    om::M3 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = relo_r[3 * i + j];
    ((Pipeline *)h)->est.setReloFrame(stamp, index, v, V3(relo_t[0], relo_t[1], relo_t[2]), R);
}
// what pubRelocalization / the pose graph read (visualization.cpp:454-538): out30 = relo_relative_t(3) relo_relative_q(w x y z)
// relo_relative_yaw drift_correct_t(3) drift_correct_r(9 row-major) relo_Pose(7) relocalization_info relo_frame_local_index n_relo_factors
void ovio_get_relo(void *h, double *out30) {
    const Estimator &e = ((Pipeline *)h)->est;
    double *o = out30;
    *o++ = e.relo_relative_t.x; *o++ = e.relo_relative_t.y; *o++ = e.relo_relative_t.z;
    *o++ = e.relo_relative_q.w; *o++ = e.relo_relative_q.x; *o++ = e.relo_relative_q.y; *o++ = e.relo_relative_q.z;
    *o++ = e.relo_relative_yaw;
    *o++ = e.drift_correct_t.x; *o++ = e.drift_correct_t.y; *o++ = e.drift_correct_t.z;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) *o++ = e.drift_correct_r(i, j);
    for (int k = 0; k < 7; k++) *o++ = e.relo_Pose[k];
    *o++ = e.relocalization_info ? 1 : 0; *o++ = e.relo_frame_local_index; *o++ = e.relo_residuals;
}
// colour / depth pairing (estimator_nodelet.cpp:200-232): pairs[2 k] / [2 k + 1] = colour / depth index; thrown2 = dropped colour, depth
int ovio_pair_color_depth(int nc, const double *tc, int nd, const double *td, int *pairs, int *thrown2) {
    auto p = pair_color_depth(std::vector<double>(tc, tc + nc), std::vector<double>(td, td + nd), thrown2);
    for (size_t k = 0; k < p.size(); k++) { pairs[2 * k] = p[k].first; pairs[2 * k + 1] = p[k].second; }
    return (int)p.size();
}
void ovio_push_imu(void *h, double t, const double *acc, const double *gyr) {
    ((Pipeline *)h)->est.inputIMU(t, V3(acc[0], acc[1], acc[2]), V3(gyr[0], gyr[1], gyr[2]));
}
void ovio_push_imu_n(void *h, int n, const double *t, const double *acc, const double *gyr) {
    for (int i = 0; i < n; i++)
        ((Pipeline *)h)->est.inputIMU(t[i], V3(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]), V3(gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]));
}
int ovio_feed(void *h, const uint8_t *gray, const uint16_t *depth, double t) { return ((Pipeline *)h)->feed(gray, depth, t); }
int ovio_feed_mode(void *h, const uint8_t *gray, const uint16_t *depth, double t, int mode) { return ((Pipeline *)h)->feed(gray, depth, t, mode); }
// the tracker half: returns the number of packaged features (0 = nothing for processImage); ids ascending, obs 7 doubles each
int ovio_track(void *h, const uint8_t *gray, double t, int mode, const double *R_in, int cap, int *ids, double *obs) {
    std::map<int, std::array<double, 7>> image;
    if (!((Pipeline *)h)->track(gray, t, mode, R_in, image)) return 0;
    int n = 0;
    for (auto &kv : image) {
        if (n >= cap) break;
        ids[n] = kv.first;
        for (int k = 0; k < 7; k++) obs[7 * n + k] = kv.second[k];
        n++;
    }
    return (int)image.size();
}
// the estimator half on a caller-supplied map: returns 1 processed, 0 need-IMU (nothing consumed)
int ovio_process_obs(void *h, int n, const int *ids, const double *obs, const uint16_t *depth, double t) {
    std::map<int, std::array<double, 7>> image;
    for (int i = 0; i < n; i++) { std::array<double, 7> a; for (int k = 0; k < 7; k++) a[k] = obs[7 * i + k]; image[ids[i]] = a; }
    return ((Pipeline *)h)->process(image, depth, t);
}
void ovio_set_fisheye_mask(void *h, const uint8_t *mask) {   // NULL = off
    Pipeline *p = (Pipeline *)h;
    if (mask) p->tracker.fisheye_mask.assign(mask, mask + (size_t)p->tracker.cfg.width * p->tracker.cfg.height);
    else p->tracker.fisheye_mask.clear();
}
void ovio_set_tracker_lag(void *h, int lag) { ((Pipeline *)h)->tracker_lag = lag; }
void ovio_latest_odometry(void *h, double *out11) { ((Pipeline *)h)->est.latestOdometry(out11); }
void ovio_predict_motion(void *h, double t0, double t1, double *R9) { ((Pipeline *)h)->est.predictMotion(t0, t1, R9); }
void *ovio_gate_create(int freq, int frontend_freq) { return new FrameGate(freq, frontend_freq); }
void ovio_gate_destroy(void *g) { delete (FrameGate *)g; }
int ovio_gate_step(void *g, double t) { return ((FrameGate *)g)->step(t); }
void ovio_gate_empty_map(void *g, double t) { ((FrameGate *)g)->empty_map(t); }

// out: [solver_flag, frame_count, marginalization_flag, td, n_landmarks, last_track_num, reboot_count, frames_processed,
//       iterations, successful, initial_cost, final_cost, n_lm_in_problem, n_residuals, n_var_landmarks, has_prior]
void ovio_get_status(void *h, double *out) {
    Pipeline *p = (Pipeline *)h;
    Estimator &e = p->est;
    out[0] = e.solver_flag; out[1] = e.frame_count; out[2] = e.marginalization_flag; out[3] = e.td;
    out[4] = (double)e.feature.size(); out[5] = e.last_track_num; out[6] = e.reboot_count; out[7] = p->frames_processed;
    out[8] = e.last_stats.iterations; out[9] = e.last_stats.successful; out[10] = e.last_stats.initial_cost;
    out[11] = e.last_stats.final_cost; out[12] = e.last_stats.n_landmarks; out[13] = e.last_stats.n_residuals;
    out[14] = e.last_stats.n_var_landmarks; out[15] = e.has_prior;
}
// out2 = (candidate steps cut by the inverse-depth upper bound, bounded landmarks that entered solves) since construction
void ovio_get_bound_stats(void *h, double *out2) {
    const Estimator &e = ((Pipeline *)h)->est;
    out2[0] = (double)e.bound_clamps; out2[1] = (double)e.bounded_landmark_solves;
}
// out2 = (trial evaluations, shortened steps) of the Armijo line search of bounds-constrained solves since construction
void ovio_get_line_search_stats(void *h, double *out2) {
    const Estimator &e = ((Pipeline *)h)->est;
    out2[0] = (double)e.line_search_evals; out2[1] = (double)e.line_search_contractions;
}
// the scalar step of the line search on its own (tests): samples = rows (x, value, gradient, valid) lower / previous / current
double ovio_ls_next_step(const double *lower, const double *previous, const double *current, double lo, double hi) {
    om::LsSample a = {lower[0], lower[1], lower[2], (int)lower[3]}, b = {previous[0], previous[1], previous[2], (int)previous[3]},
                 c = {current[0], current[1], current[2], (int)current[3]};
    double ws[96];
    return om::ls_next_step(a, b, c, lo, hi, ws);
}
int ovio_poly_roots_real(const double *c, int n, double *re) { double ws[16]; return om::ls_poly_roots_real(c, n, re, ws); }
// window arrays, each (W+1) rows: P(3) Q(wxyz 4) V(3) Ba(3) Bg(3) stamp(1) = 17 doubles per frame
void ovio_get_window(void *h, double *out) {
    Estimator &e = ((Pipeline *)h)->est;
    for (int i = 0; i <= e.W; i++) {
        double *o = out + 17 * i;
        Q q = fromR(e.Rs[i]);
        o[0] = e.Ps[i].x; o[1] = e.Ps[i].y; o[2] = e.Ps[i].z;
        o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
        o[7] = e.Vs[i].x; o[8] = e.Vs[i].y; o[9] = e.Vs[i].z;
        o[10] = e.Bas[i].x; o[11] = e.Bas[i].y; o[12] = e.Bas[i].z;
        o[13] = e.Bgs[i].x; o[14] = e.Bgs[i].y; o[15] = e.Bgs[i].z;
        o[16] = e.Headers[i];
    }
}
void ovio_get_extrinsic(void *h, double *out) {  // tic(3) + ric row-major (9) + td
    Estimator &e = ((Pipeline *)h)->est;
    out[0] = e.tic.x; out[1] = e.tic.y; out[2] = e.tic.z;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[3 + i * 3 + j] = e.ric(i, j);
    out[12] = e.td;
}
// landmark table: per landmark [id, start_frame, n_obs, estimated_depth, estimate_flag, solve_flag, is_dynamic]
int ovio_get_landmarks(void *h, int cap, double *out) {
    Estimator &e = ((Pipeline *)h)->est;
    int n = 0;
    for (auto &l : e.feature) {
        if (n >= cap) break;
        double *o = out + 7 * n++;
        o[0] = l.feature_id; o[1] = l.start_frame; o[2] = (double)l.obs.size(); o[3] = l.estimated_depth;
        o[4] = l.estimate_flag; o[5] = l.solve_flag; o[6] = l.is_dynamic;
    }
    return (int)e.feature.size();
}

// 12 per landmark: the 7 above + feature_per_frame[0].point (3), feature_per_frame[0].depth, feature_per_frame.back().depth
int ovio_get_landmarks_ex(void *h, int cap, double *out) {
    Estimator &e = ((Pipeline *)h)->est;
    int n = 0;
    for (auto &l : e.feature) {
        if (n >= cap) break;
        double *o = out + 12 * n++;
        o[0] = l.feature_id; o[1] = l.start_frame; o[2] = (double)l.obs.size(); o[3] = l.estimated_depth;
        o[4] = l.estimate_flag; o[5] = l.solve_flag; o[6] = l.is_dynamic;
        o[7] = l.obs.front().x; o[8] = l.obs.front().y; o[9] = l.obs.front().z; o[10] = l.obs.front().depth; o[11] = l.obs.back().depth;
    }
    return (int)e.feature.size();
}

static int tracker_get(Tracker &t, int cap, int *ids, int *cnt, float *cur, float *un, float *vel) {
    int n = (int)t.ids.size();
    for (int i = 0; i < n && i < cap; i++) {
        ids[i] = t.ids[i]; cnt[i] = t.track_cnt[i];
        cur[2 * i] = t.cur_pts[i].x; cur[2 * i + 1] = t.cur_pts[i].y;
        un[2 * i] = t.cur_un_pts[i].x; un[2 * i + 1] = t.cur_un_pts[i].y;
        vel[2 * i] = t.pts_velocity[i].x; vel[2 * i + 1] = t.pts_velocity[i].y;
    }
    return n;
}
int ovio_get_tracks(void *h, int cap, int *ids, int *cnt, float *cur, float *un, float *vel) {
    return tracker_get(((Pipeline *)h)->tracker, cap, ids, cnt, cur, un, vel);
}

// ---------------------------------------------------------------- stand-alone tracker (front-end parity)
void *ovio_tracker_create(const Config *c) { return new Tracker(*c); }
void ovio_tracker_destroy(void *h) { delete (Tracker *)h; }
void ovio_tracker_read(void *h, const uint8_t *gray, double t, const double *R, int publish) {
    Tracker *tr = (Tracker *)h;
    tr->readImage(gray, t, R, publish != 0);
    tr->updateIDs();
}
int ovio_tracker_get(void *h, int cap, int *ids, int *cnt, float *cur, float *un, float *vel) {
    return tracker_get(*(Tracker *)h, cap, ids, cnt, cur, un, vel);
}
int ovio_tracker_grid(void *h, int *rects /*4 per cell*/, int *threshold) {
    Tracker *tr = (Tracker *)h;
    for (size_t i = 0; i < tr->grids_rect.size(); i++) {
        rects[4 * i] = tr->grids_rect[i].x; rects[4 * i + 1] = tr->grids_rect[i].y;
        rects[4 * i + 2] = tr->grids_rect[i].w; rects[4 * i + 3] = tr->grids_rect[i].h;
    }
    *threshold = tr->grids_threshold;
    return (int)tr->grids_rect.size();
}

// ---------------------------------------------------------------- primitives (known-answer tests)
void ovio_cam_lift(const Config *c, int n, const double *uv, double *xy) {
    for (int i = 0; i < n; i++) cam_lift(*c, uv[2 * i], uv[2 * i + 1], xy[2 * i], xy[2 * i + 1]);
}
void ovio_cam_project(const Config *c, int n, const double *XYZ, double *uv) {
    for (int i = 0; i < n; i++) cam_project(*c, XYZ[3 * i], XYZ[3 * i + 1], XYZ[3 * i + 2], uv[2 * i], uv[2 * i + 1]);
}
void ovio_clahe(const uint8_t *src, int w, int h, uint8_t *dst) { clahe_apply(src, w, h, dst, 3.0, 8); }
void ovio_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst) {
    Image s, d;
    s.w = w; s.h = h; s.d.assign(src, src + (size_t)w * h);
    pyr_down(s, d);
    std::memcpy(dst, d.d.data(), d.d.size());
}
int ovio_fast_score(const uint8_t *patch7x7) { return fast_corner_score(patch7x7 + 3 * 7 + 3, 7, 10); }
int ovio_fast_roi(const uint8_t *img, int W, int H, int rx, int ry, int rw, int rh, int cap, float *out /*x,y,score*/) {
    std::vector<KeyPt> k;
    fast_detect_roi(img, W, H, rx, ry, rw, rh, k);
    for (size_t i = 0; i < k.size() && (int)i < cap; i++) { out[3 * i] = k[i].x; out[3 * i + 1] = k[i].y; out[3 * i + 2] = k[i].response; }
    return (int)k.size();
}
// ---- pose_graph slice (posegraph.cpp)
// KeyFrame::computeWindowBRIEFPoint + computeBRIEFPoint (keyframe.cpp:80-124): blur once, FAST(threshold, NMS) on the RAW image, BRIEF of
// the window points and of the keypoints on the blurred image, normalised keypoints through the camera model.  Returns the keypoint count.
int ovio_pg_describe(const Config *cfg, const uint8_t *gray, int n_win, const float *win_uv, const int *pattern1024, int fast_threshold,
                     uint64_t *win_desc, int cap, float *kp_xy, uint64_t *kp_desc, float *kp_norm) {
    const int W = cfg->width, H = cfg->height;
    std::vector<uint8_t> blur((size_t)W * H);
    gaussian_blur_9x9(gray, W, H, blur.data());
    if (n_win > 0) brief_compute(blur.data(), W, H, win_uv, n_win, pattern1024, win_desc);
    std::vector<KeyPt> k;
    fast_detect_roi(gray, W, H, 0, 0, W, H, k, fast_threshold);
    const int m = std::min((int)k.size(), cap);
    std::vector<float> xy(2 * (size_t)std::max(m, 1));
    for (int i = 0; i < m; i++) {
        xy[2 * i] = k[i].x; xy[2 * i + 1] = k[i].y;
        kp_xy[2 * i] = k[i].x; kp_xy[2 * i + 1] = k[i].y;
        double x, y;
        cam_lift(*cfg, k[i].x, k[i].y, x, y);
        kp_norm[2 * i] = (float)x; kp_norm[2 * i + 1] = (float)y;
    }
    if (m > 0) brief_compute(blur.data(), W, H, xy.data(), m, pattern1024, kp_desc);
    return (int)k.size();
}
// ---- place recognition (bow.cpp)
void *ovio_bow_load(const char *path) { BowVoc *v = new BowVoc(); if (!v->load_bin(path)) { delete v; return nullptr; } return v; }
void *ovio_bow_create(int k, int L, int scoring, int weighting, int nn, const int32_t *nid, const int32_t *pid, const double *w, const uint64_t *d, int nw,
                      const int32_t *wn, const int32_t *wi) {
    BowVoc *v = new BowVoc();
    if (!v->build(k, L, scoring, weighting, nn, nid, pid, w, d, nw, wn, wi)) { delete v; return nullptr; }
    return v;
}
void ovio_bow_destroy(void *h) { delete (BowVoc *)h; }
void ovio_bow_info(void *h, int *out6) {
    BowVoc *v = (BowVoc *)h;
    out6[0] = v->k; out6[1] = v->L; out6[2] = v->scoring; out6[3] = v->weighting; out6[4] = (int)v->nodes.size() - 1; out6[5] = (int)v->words.size();
}
void ovio_bow_transform(void *h, const uint64_t *desc, int n, int *word_id, double *weight) {
    for (int i = 0; i < n; i++) ((BowVoc *)h)->transform_one(desc + (size_t)i * 4, word_id[i], weight[i]);
}
int ovio_bow_vector(void *h, const uint64_t *desc, int n, int cap, int *word_id, double *value) {
    std::map<int, double> v;
    ((BowVoc *)h)->transform(desc, n, v);
    int m = 0;
    for (auto &e : v) { if (m < cap) { word_id[m] = e.first; value[m] = e.second; } m++; }
    return m;
}
int ovio_bow_add(void *h, const uint64_t *desc, int n) { return ((BowVoc *)h)->add(desc, n); }
int ovio_bow_query(void *h, const uint64_t *desc, int n, int max_results, int max_id, int *ids, double *scores) {
    std::vector<std::pair<int, double>> ret;
    ((BowVoc *)h)->query(desc, n, max_results, max_id, ret);
    for (size_t i = 0; i < ret.size(); i++) { ids[i] = ret[i].first; scores[i] = ret[i].second; }
    return (int)ret.size();
}
int ovio_bow_detect_loop(void *h, const uint64_t *desc, int n, int frame_index) { return ((BowVoc *)h)->detect_loop(desc, n, frame_index); }
void ovio_pg_blur(const uint8_t *gray, int W, int H, uint8_t *out) { gaussian_blur_9x9(gray, W, H, out); }
void ovio_pg_match(const uint64_t *wd, int n, const uint64_t *od, int m, int *best_index, int *best_dist) { brief_match(wd, n, od, m, best_index, best_dist); }
int ovio_pg_find_connection(int n, const float *pt3d, const float *pt_norm, const double *pt_id, const int *match, const float *old_norm,
                            const double *vio_T, const double *vio_R, const double *qic9, const double *tic3, int min_loop_num,
                            double *loop_info8, double *match_points, int *n_match_out, double *pnp_T3, double *pnp_R9) {
    return find_connection(n, pt3d, pt_norm, pt_id, match, old_norm, vio_T, vio_R, qic9, tic3, min_loop_num, loop_info8, match_points, n_match_out, pnp_T3, pnp_R9);
}
void ovio_pg_optimize6dof(int n, const double *t_in, const double *R_in, const int *sequence, const int *loop_to, const double *loop_info,
                          double *t_out, double *R_out, double *drift12) {
    optimize_6dof(n, t_in, R_in, sequence, loop_to, loop_info, t_out, R_out, drift12);
}
void ovio_pg_optimize4dof(int n, const double *t_in, const double *R_in, const int *sequence, const int *loop_to, const double *loop_info,
                          double *t_out, double *R_out, double *drift4) {
    optimize_4dof(n, t_in, R_in, sequence, loop_to, loop_info, t_out, R_out, drift4);
}
void ovio_circle_hw(int radius, int *hw) {
    std::vector<int> v;
    circle_halfwidths(radius, v);
    for (int i = 0; i <= radius; i++) hw[i] = v[i];
}
void ovio_lk(const uint8_t *prev, const uint8_t *next, int w, int h, int maxLevel, int n, const float *prevPts, float *nextPts,
             uint8_t *status, int useInitial) {
    std::vector<Image> P(maxLevel + 1), N(maxLevel + 1);
    P[0].w = N[0].w = w; P[0].h = N[0].h = h;
    P[0].d.assign(prev, prev + (size_t)w * h);
    N[0].d.assign(next, next + (size_t)w * h);
    for (int l = 1; l <= maxLevel; l++) { pyr_down(P[l - 1], P[l]); pyr_down(N[l - 1], N[l]); }
    std::vector<P2f> pp(n), np(n);
    for (int i = 0; i < n; i++) { pp[i] = P2f{prevPts[2 * i], prevPts[2 * i + 1]}; np[i] = P2f{nextPts[2 * i], nextPts[2 * i + 1]}; }
    std::vector<uint8_t> st;
    lk_track(P, N, pp, np, st, maxLevel, useInitial != 0);
    for (int i = 0; i < n; i++) { nextPts[2 * i] = np[i].x; nextPts[2 * i + 1] = np[i].y; status[i] = st[i]; }
}
// 7 correspondences (normalised coordinates) -> up to 3 fundamental matrices (row-major 9 each); returns their number
int ovio_seven_point(const double *x1, const double *y1, const double *x2, const double *y2, double *F27) {
    double F[3][9];
    int n = seven_point_models(x1, y1, x2, y2, F);
    for (int k = 0; k < n; k++) for (int i = 0; i < 9; i++) F27[9 * k + i] = F[k][i];
    return n;
}
void ovio_ransac(const Config *c, int n, const float *p1, const float *p2, uint8_t *status) {
    std::vector<P2f> a(n), b(n);
    for (int i = 0; i < n; i++) { a[i] = P2f{p1[2 * i], p1[2 * i + 1]}; b[i] = P2f{p2[2 * i], p2[2 * i + 1]}; }
    std::vector<uint8_t> st;
    ransac_fundamental(*c, a, b, st);
    for (int i = 0; i < n; i++) status[i] = st[i];
}
// obs: 9 doubles each (x,y,z,u,v,vx,vy,cur_td,depth). J out: Ji(14) Jj(14) Jex(14) Jl(2) Jtd(2)
void ovio_eval_projection(const Config *c, const double *pi, const double *pj, const double *ex, double inv_dep, double td,
                          const double *oi, const double *oj, int use_td, double *r, double *J) {
    Obs a{oi[0], oi[1], oi[2], oi[3], oi[4], oi[5], oi[6], oi[7], oi[8]}, b{oj[0], oj[1], oj[2], oj[3], oj[4], oj[5], oj[6], oj[7], oj[8]};
    eval_projection(*c, pi, pj, ex, inv_dep, td, a, b, use_td != 0, r, J ? J : nullptr, J ? J + 14 : nullptr, J ? J + 28 : nullptr,
                    J ? J + 42 : nullptr, J ? J + 44 : nullptr);
}
// pre-integration: n samples (dt, acc, gyr); first acc0/gyr0; out: delta_p(3) delta_q(wxyz) delta_v(3) sum_dt jac(225) cov(225)
void *ovio_preint_create(const Config *c, const double *acc0, const double *gyr0, const double *ba, const double *bg) {
    return new Integration(*c, V3(acc0[0], acc0[1], acc0[2]), V3(gyr0[0], gyr0[1], gyr0[2]), V3(ba[0], ba[1], ba[2]), V3(bg[0], bg[1], bg[2]));
}
void ovio_preint_destroy(void *h) { delete (Integration *)h; }
void ovio_preint_push(void *h, double dt, const double *acc, const double *gyr) {
    ((Integration *)h)->push_back(dt, V3(acc[0], acc[1], acc[2]), V3(gyr[0], gyr[1], gyr[2]));
}
void ovio_preint_repropagate(void *h, const double *ba, const double *bg) {
    ((Integration *)h)->repropagate(V3(ba[0], ba[1], ba[2]), V3(bg[0], bg[1], bg[2]));
}
void ovio_preint_get(void *h, double *out) {
    Integration *p = (Integration *)h;
    out[0] = p->delta_p.x; out[1] = p->delta_p.y; out[2] = p->delta_p.z;
    out[3] = p->delta_q.w; out[4] = p->delta_q.x; out[5] = p->delta_q.y; out[6] = p->delta_q.z;
    out[7] = p->delta_v.x; out[8] = p->delta_v.y; out[9] = p->delta_v.z;
    out[10] = p->sum_dt;
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) { out[11 + i * 15 + j] = p->jacobian[i][j]; out[11 + 225 + i * 15 + j] = p->covariance[i][j]; }
}
// r(15), J: Ji(105) Jsi(135) Jj(105) Jsj(135)
void ovio_eval_imu(void *h, double g_norm, const double *pi, const double *sbi, const double *pj, const double *sbj, double *r, double *J) {
    eval_imu(*(Integration *)h, V3(0, 0, g_norm), pi, sbi, pj, sbj, r, J ? J : nullptr, J ? J + 105 : nullptr, J ? J + 240 : nullptr,
             J ? J + 345 : nullptr);
}
// symmetric eigen (for tests of the stand-in solver)
void ovio_sym_eig(int n, const double *A, double *w, double *V) {
    Mat M(n, n), Vm;
    for (int i = 0; i < n * n; i++) M.d[i] = A[i];
    std::vector<double> ww;
    sym_eig(M, ww, Vm);
    for (int i = 0; i < n; i++) w[i] = ww[i];
    for (int i = 0; i < n * n; i++) V[i] = Vm.d[i];
}
// MarginalizationInfo::marginalize() tail (marginalization_factor.cpp:262-315) on a caller-supplied A (N×N, N = m+n), b:
// J (n×n row-major), r (n)
void ovio_marg_finish(int m, int n, const double *A, const double *b, double *J, double *r) {
    Config c;
    Estimator e(c);
    Mat M(m + n, m + n);
    for (int i = 0; i < (m + n) * (m + n); i++) M.d[i] = A[i];
    std::vector<double> bb(b, b + m + n);
    marg_finish(e, M, bb, m, n);
    const bool quad = (e.deviations & ODEV_QUADRATIC_PRIOR) != 0;   // attribution variant (OVIO_DEVIATIONS): (A, b) come back instead of (J, r)
    for (int i = 0; i < n * n; i++) J[i] = quad ? e.prior_A.d[i] : e.prior_J.d[i];
    for (int i = 0; i < n; i++) r[i] = quad ? e.prior_b[i] : e.prior_r[i];
}
// sincos_det (om.h): n angles -> sin, cos.  And the attribution experiment's switch for the free factor functions (oracle.h ODEV_*): returns the
// previous mask; pipelines read OVIO_DEVIATIONS themselves when they are created
void ovio_sincos_det(int n, const double *x, double *sn, double *cs) { for (int i = 0; i < n; i++) sincos_det(x[i], &sn[i], &cs[i]); }
int ovio_set_deviations(int mask) { int old = oracle_deviations; oracle_deviations = mask; return old; }
// prior accessors (marginalisation tests): returns n; J (n×n row-major), r (n), present (W+3)
int ovio_get_prior(void *h, double *J, double *r, double *x0, uint8_t *present) {
    Estimator &e = ((Pipeline *)h)->est;
    if (!e.has_prior) return 0;
    int n = e.prior_n;
    const bool quad = (e.deviations & ODEV_QUADRATIC_PRIOR) != 0;   // attribution variant: (A, b) instead of (J, r)
    if (J) for (int i = 0; i < n * n; i++) J[i] = quad ? e.prior_A.d[i] : e.prior_J.d[i];
    if (r) for (int i = 0; i < n; i++) r[i] = quad ? e.prior_b[i] : e.prior_r[i];
    if (x0) for (size_t i = 0; i < e.prior_x0.size(); i++) x0[i] = e.prior_x0[i];
    if (present) for (size_t i = 0; i < e.prior_present.size(); i++) present[i] = e.prior_present[i];
    return n;
}

// ---------------------------------------------------------------- visual-inertial alignment (dynamic initialisation, part)
// frames: n x {R[9] row-major, T[3], sum_dt, delta_p[3], delta_v[3]} = 19 doubles each.  x_out: 3 n + 2 doubles (body velocities
// per frame, then the last tangent-plane correction).  Returns 1 on success (|g| within 1 m/s^2 of g_norm before refinement).
int ovio_linear_alignment_with_depth(int n, const double *frames, const double *tic, double g_norm, double *g_out, double *x_out) {
    std::vector<AlignFrame> f(n);
    for (int i = 0; i < n; i++) {
        const double *p = frames + 19 * i;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) f[i].R(r, c) = p[3 * r + c];
        f[i].T = V3(p[9], p[10], p[11]);
        f[i].sum_dt = p[12];
        f[i].delta_p = V3(p[13], p[14], p[15]);
        f[i].delta_v = V3(p[16], p[17], p[18]);
    }
    V3 g;
    std::vector<double> x;
    bool ok = linear_alignment_with_depth(f, V3(tic[0], tic[1], tic[2]), g_norm, g, x);
    g_out[0] = g.x; g_out[1] = g.y; g_out[2] = g.z;
    for (size_t i = 0; i < x.size() && i < (size_t)(3 * n + 3); i++) x_out[i] = x[i];
    return ok ? 1 : 0;
}
// obj[n*3], img[n*2]; R[9] row-major / t[3] in (guess) and out; camera_point = R X + t
int ovio_solve_pnp_iterative(int n, const double *obj, const double *img, double *R, double *t) {
    std::vector<V3> o(n);
    std::vector<std::array<double, 2>> im(n);
    for (int i = 0; i < n; i++) { o[i] = V3(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]); im[i] = {img[2 * i], img[2 * i + 1]}; }
    M3 Rm;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rm(r, c) = R[3 * r + c];
    V3 tv(t[0], t[1], t[2]);
    bool ok = solve_pnp_iterative(o, im, Rm, tv);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = Rm(r, c);
    t[0] = tv.x; t[1] = tv.y; t[2] = tv.z;
    return ok ? 1 : 0;
}
int ovio_solve_pnp_ransac_epnp(int n, const double *obj, const double *img, int max_iters, double thresh, double confidence, double *R,
                               double *t, uint8_t *inliers) {
    std::vector<V3> o(n);
    std::vector<std::array<double, 2>> im(n);
    for (int i = 0; i < n; i++) { o[i] = V3(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]); im[i] = {img[2 * i], img[2 * i + 1]}; }
    M3 Rm;
    V3 tv;
    std::vector<uint8_t> in;
    bool ok = solve_pnp_ransac_epnp(o, im, max_iters, thresh, confidence, Rm, tv, in);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = Rm(r, c);
    t[0] = tv.x; t[1] = tv.y; t[2] = tv.z;
    for (int i = 0; i < n && i < (int)in.size(); i++) inliers[i] = in[i];
    return ok ? 1 : 0;
}
// Vision-only structure of the initialisation window: relativePose + GlobalSFM::construct.
// Features: start[nf], nobs[nf], obs = concatenated (x, y, depth) per observation (consecutive frames from start).
// Out: l, q[frame_num*4] (w,x,y,z) and T[frame_num*3] = camera poses in the frame of camera l, pts[nf*4] = (state, X, Y, Z),
// stats[4] = (BA iterations, initial cost, final cost, converged).  Returns 0 ok, 1 no frame pair with enough parallax, 2 SfM failed.
int ovio_sfm_window(int window_size, int nf, const int *start, const int *nobs, const double *obs, int *l_out, double *q_out,
                    double *T_out, double *pts_out, double *stats_out) {
    std::vector<SfmFeature> f(nf);
    size_t off = 0;
    for (int i = 0; i < nf; i++) {
        f[i].id = i;
        for (int k = 0; k < nobs[i]; k++, off++) {
            f[i].observation.push_back({start[i] + k, {obs[3 * off], obs[3 * off + 1]}});
            f[i].observation_depth.push_back({start[i] + k, obs[3 * off + 2]});
        }
    }
    M3 rR;
    V3 rT;
    int l = -1;
    if (!sfm_relative_pose(window_size, f, rR, rT, l)) return 1;
    *l_out = l;
    const int fn = window_size + 1;
    std::vector<Q> q(fn);
    std::vector<V3> Tv(fn);
    std::map<int, V3> tracked;
    SfmStats st;
    if (!sfm_construct(fn, q.data(), Tv.data(), l, rR, rT, f, tracked, &st)) return 2;
    for (int i = 0; i < fn; i++) {
        q_out[4 * i] = q[i].w; q_out[4 * i + 1] = q[i].x; q_out[4 * i + 2] = q[i].y; q_out[4 * i + 3] = q[i].z;
        T_out[3 * i] = Tv[i].x; T_out[3 * i + 1] = Tv[i].y; T_out[3 * i + 2] = Tv[i].z;
    }
    for (int i = 0; i < nf; i++) { pts_out[4 * i] = f[i].state; for (int k = 0; k < 3; k++) pts_out[4 * i + 1 + k] = f[i].position[k]; }
    stats_out[0] = st.iterations; stats_out[1] = st.initial_cost; stats_out[2] = st.final_cost; stats_out[3] = st.converged;
    return 0;
}
void ovio_tangent_basis(const double *g0, double *b, double *c) {
    V3 bb, cc;
    tangent_basis(V3(g0[0], g0[1], g0[2]), bb, cc);
    for (int k = 0; k < 3; k++) { b[k] = bb[k]; c[k] = cc[k]; }
}
// in/out: Ps[n*3] (camera positions in the SfM frame -> body positions in the gravity-aligned world), Rs[n*9], out Vs[n*3]; g in/out
void ovio_align_window_to_gravity(int n, double *Ps, double *Rs, double *Vs, const double *x, const double *tic, double *g) {
    std::vector<V3> P(n), Vv(n);
    std::vector<M3> R(n);
    for (int i = 0; i < n; i++) {
        P[i] = V3(Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2]);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[i](r, c) = Rs[9 * i + 3 * r + c];
    }
    std::vector<double> xv(x, x + 3 * n);
    V3 gv(g[0], g[1], g[2]);
    align_window_to_gravity(n, P.data(), R.data(), Vv.data(), xv, V3(tic[0], tic[1], tic[2]), gv);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) { Ps[3 * i + k] = P[i][k]; Vs[3 * i + k] = Vv[i][k]; }
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rs[9 * i + 3 * r + c] = R[i](r, c);
    }
    g[0] = gv.x; g[1] = gv.y; g[2] = gv.z;
}

}  // extern "C"
