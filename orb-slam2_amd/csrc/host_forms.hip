// Host-pointer forms of the entry points a C++ caller with host-resident data needs besides the
// windowed / projection matchers: ORBmatcher::SearchByBoW (src/ORBmatcher.cc:158-295, :726-861),
// ORBmatcher::SearchForTriangulation (:863-1025),
// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:191-256), Frame::UndistortKeyPoints
// (src/Frame.cc:290-320) and the per-feature vocabulary descent behind Frame::ComputeBoW (:281-288).
// and the two "map points in, matches out" forms behind the projection matchers, Fuse and SearchBySim3
// (orbhip_project_and_search, orbhip_project_and_best_in_radius: probes are built and consumed on the
// device).  Each copies its arguments into a staging arena of the handle, runs the device entry points on
// the matcher's stream and copies the results back (blocking).  No arithmetic of the path happens here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/orbhip.h"
#include "common.h"
#include "matcher_internal.h"

namespace {

// Carves 256-byte aligned pieces out of the handle's staging arena; reserve() grows the arena once all
// pieces are known.
struct Arena {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

int reserve_stage(orbhip_matcher* m, size_t bytes) {
    if (bytes <= m->stage_bytes) return ORBHIP_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->d_stage) (void)hipFree(m->d_stage);
    m->d_stage = nullptr;
    m->stage_bytes = 0;
    if (hipMalloc(&m->d_stage, bytes) != hipSuccess) return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", bytes);
    m->stage_bytes = bytes;
    return ORBHIP_OK;
}

struct FvHost {
    const int32_t *nodes, *start, *feat;
    int count;
};

// A FeatureVector handed over as arrays must be what FeatureVector::addFeature builds: nodes strictly
// ascending, starts non-decreasing from 0, feature indices inside the frame.
int check_fv(const FvHost& fv, int n, const char* which) {
    if (fv.count < 0) return orb_fail(ORBHIP_EINVAL, "%s: negative node count", which);
    if (fv.count == 0) return ORBHIP_OK;
    if (!fv.nodes || !fv.start || !fv.feat) return orb_fail(ORBHIP_EINVAL, "%s: null array", which);
    if (fv.start[0] != 0) return orb_fail(ORBHIP_EINVAL, "%s: start[0] = %d", which, fv.start[0]);
    for (int j = 0; j < fv.count; j++) {
        if (j && fv.nodes[j] <= fv.nodes[j - 1]) return orb_fail(ORBHIP_EINVAL, "%s: node ids not ascending at %d", which, j);
        if (fv.start[j + 1] < fv.start[j]) return orb_fail(ORBHIP_EINVAL, "%s: starts decrease at %d", which, j);
    }
    const int total = fv.start[fv.count];
    if (total > n) return orb_fail(ORBHIP_EINVAL, "%s: %d feature entries for %d features", which, total, n);
    for (int i = 0; i < total; i++)
        if (fv.feat[i] < 0 || fv.feat[i] >= n) return orb_fail(ORBHIP_EINVAL, "%s: feature index %d of %d", which, fv.feat[i], n);
    return ORBHIP_OK;
}

// Stages one target frame and one point set, builds the frame's grid and the probes of the points on the device.
struct ProjectStage {
    char* b = nullptr;
    size_t o_pr, o_pd, o_kp, o_d, o_t, o_n, o_cs, o_ci, o_a, o_b, o_scalar;
    int cap, pcap;
};

int project_stage(orbhip_matcher* m, int kind, const OrbProjection* cam, const OrbProjection* cam_to, const float* scale_factors,
                  int nlevels, const OrbPointSet* pts, float th, float viewing_cos_limit, const OrbKeypoint* kp2,
                  const uint8_t* desc2, int n2, OrbFrameBounds bounds, const uint8_t* taken, size_t out_a_bytes_per,
                  bool out_per_probe, ProjectStage& st) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (!pts || !cam || !scale_factors) return orb_fail(ORBHIP_EINVAL, "null argument");
    const int np = pts->n;
    if (np < 0 || n2 < 0 || nlevels < 1 || nlevels > 64) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (np && (!pts->pos || !pts->desc)) return orb_fail(ORBHIP_EINVAL, "a point set needs positions and descriptors");
    if (n2 && (!kp2 || !desc2)) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if ((kind == ORBHIP_PROBES_SIM3) != (cam_to != nullptr))
        return orb_fail(ORBHIP_EINVAL, "a second camera goes with ORBHIP_PROBES_SIM3 and with no other kind");
    const int cap = std::max(1, n2), pcap = std::max(1, np);
    if (cap > ORBHIP_FRAME_MAX_KEYPOINTS) return orb_fail(ORBHIP_ECAP, "%d features > %d per frame", cap, ORBHIP_FRAME_MAX_KEYPOINTS);
    HIPCHK(hipSetDevice(m->device));
    Arena a;
    st.cap = cap;
    st.pcap = pcap;
    st.o_pr = a.take(sizeof(OrbProbe) * pcap);
    st.o_pd = a.take(32 * (size_t)pcap);
    st.o_kp = a.take(sizeof(OrbKeypoint) * cap);
    st.o_d = a.take(32 * (size_t)cap);
    st.o_t = a.take(cap);
    st.o_n = a.take(sizeof(int32_t) * 2);
    st.o_cs = a.take(sizeof(int32_t) * (ORBHIP_FRAME_GRID_CELLS + 1));
    st.o_ci = a.take(sizeof(int32_t) * cap);
    st.o_a = a.take(out_a_bytes_per * (out_per_probe ? pcap : cap));
    st.o_b = a.take(sizeof(int32_t) * (out_per_probe ? pcap : 1));
    const size_t o_cam = a.take(sizeof(OrbProjection) * 2), o_sf = a.take(sizeof(float) * nlevels);
    const size_t o_pos = a.take(sizeof(float) * 3 * pcap), o_nrm = a.take(sizeof(float) * 3 * pcap);
    const size_t o_min = a.take(sizeof(float) * pcap), o_max = a.take(sizeof(float) * pcap);
    const size_t o_src = a.take(sizeof(OrbKeypoint) * pcap), o_skip = a.take(pcap);
    int rc = reserve_stage(m, a.off);
    if (rc) return rc;
    char* b = st.b = (char*)m->d_stage;
    const int32_t n[2] = {np, n2};
    OrbProjection cams[2] = {*cam, cam_to ? *cam_to : *cam};
    HIPCHK(hipMemcpyAsync(b + st.o_n, n, sizeof(n), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(b + o_cam, cams, sizeof(cams), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(b + o_sf, scale_factors, sizeof(float) * nlevels, hipMemcpyHostToDevice, m->stream));
    if (np) {
        HIPCHK(hipMemcpyAsync(b + o_pos, pts->pos, sizeof(float) * 3 * np, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(b + st.o_pd, pts->desc, 32 * (size_t)np, hipMemcpyHostToDevice, m->stream));
        if (pts->normal) HIPCHK(hipMemcpyAsync(b + o_nrm, pts->normal, sizeof(float) * 3 * np, hipMemcpyHostToDevice, m->stream));
        if (pts->min_dist) HIPCHK(hipMemcpyAsync(b + o_min, pts->min_dist, sizeof(float) * np, hipMemcpyHostToDevice, m->stream));
        if (pts->max_dist) HIPCHK(hipMemcpyAsync(b + o_max, pts->max_dist, sizeof(float) * np, hipMemcpyHostToDevice, m->stream));
        if (pts->src_kp) HIPCHK(hipMemcpyAsync(b + o_src, pts->src_kp, sizeof(OrbKeypoint) * np, hipMemcpyHostToDevice, m->stream));
        if (pts->skip) HIPCHK(hipMemcpyAsync(b + o_skip, pts->skip, np, hipMemcpyHostToDevice, m->stream));
    }
    if (n2) {
        HIPCHK(hipMemcpyAsync(b + st.o_kp, kp2, sizeof(OrbKeypoint) * n2, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(b + st.o_d, desc2, 32 * (size_t)n2, hipMemcpyHostToDevice, m->stream));
        if (taken) HIPCHK(hipMemcpyAsync(b + st.o_t, taken, n2, hipMemcpyHostToDevice, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));  // n[] and cams[] live on this stack frame
    if ((rc = orbhip_frame_grid_device(m, (const OrbKeypoint*)(b + st.o_kp), (const int32_t*)(b + st.o_n) + 1, 1, cap, bounds,
                                       (int32_t*)(b + st.o_cs), (int32_t*)(b + st.o_ci))))
        return rc;
    const OrbProjection* d_cams = (const OrbProjection*)(b + o_cam);
    const float* d_sf = (const float*)(b + o_sf);
    const float* d_pos = (const float*)(b + o_pos);
    const float* d_min = pts->min_dist ? (const float*)(b + o_min) : nullptr;
    const float* d_max = pts->max_dist ? (const float*)(b + o_max) : nullptr;
    const uint8_t* d_skip = pts->skip ? (const uint8_t*)(b + o_skip) : nullptr;
    if (kind == ORBHIP_PROBES_SIM3)
        return orbhip_build_probes_sim3_device(m, d_cams, d_cams + 1, bounds, d_sf, nlevels, d_pos, d_min, d_max, d_skip,
                                               (const int32_t*)(b + st.o_n), 1, pcap, th, (OrbProbe*)(b + st.o_pr));
    return orbhip_build_probes_device(m, kind, d_cams, bounds, d_sf, nlevels, d_pos, pts->normal ? (const float*)(b + o_nrm) : nullptr,
                                      d_min, d_max, pts->src_kp ? (const OrbKeypoint*)(b + o_src) : nullptr, d_skip,
                                      (const int32_t*)(b + st.o_n), 1, pcap, th, viewing_cos_limit, (OrbProbe*)(b + st.o_pr));
}

}  // namespace

extern "C" {

int orbhip_project_and_search(orbhip_matcher* m, int kind, const OrbProjection* cam, const float* scale_factors, int nlevels,
                              const OrbPointSet* pts, float th, float viewing_cos_limit, const OrbKeypoint* kp2,
                              const uint8_t* desc2, int n2, OrbFrameBounds bounds, const uint8_t* taken, int rule, float nnratio,
                              int th_dist, int check_orientation, int32_t* match_of_feature, int32_t* nmatches,
                              OrbProbe* probes_out) {
    if (!nmatches || (n2 > 0 && !match_of_feature)) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (kind == ORBHIP_PROBES_SIM3) return orb_fail(ORBHIP_EINVAL, "ORBHIP_PROBES_SIM3 feeds orbhip_project_and_best_in_radius");
    if (pts && pts->n > ORBHIP_MAX_PROBES) return orb_fail(ORBHIP_ECAP, "%d probes > %d", pts->n, ORBHIP_MAX_PROBES);
    ProjectStage st;
    int rc = project_stage(m, kind, cam, nullptr, scale_factors, nlevels, pts, th, viewing_cos_limit, kp2, desc2, n2, bounds, taken,
                           sizeof(int32_t), false, st);
    if (rc) return rc;
    char* b = st.b;
    if ((rc = orbhip_search_by_projection_device(
             m, (const OrbProbe*)(b + st.o_pr), (const uint8_t*)(b + st.o_pd), (const int32_t*)(b + st.o_n), st.pcap,
             (const OrbKeypoint*)(b + st.o_kp), (const uint8_t*)(b + st.o_d), (const int32_t*)(b + st.o_n) + 1,
             (const int32_t*)(b + st.o_cs), (const int32_t*)(b + st.o_ci), 1, st.cap, bounds, taken ? (const uint8_t*)(b + st.o_t) : nullptr,
             rule, nnratio, th_dist, check_orientation, (int32_t*)(b + st.o_a), (int32_t*)(b + st.o_b))))
        return rc;
    if (n2) HIPCHK(hipMemcpyAsync(match_of_feature, b + st.o_a, sizeof(int32_t) * n2, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(nmatches, b + st.o_b, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    if (probes_out && pts->n) HIPCHK(hipMemcpyAsync(probes_out, b + st.o_pr, sizeof(OrbProbe) * pts->n, hipMemcpyDeviceToHost, m->stream));
    return orbhip_matcher_synchronize(m);
}

int orbhip_project_and_best_in_radius(orbhip_matcher* m, int kind, const OrbProjection* cam, const OrbProjection* cam_to,
                                      const float* scale_factors, int nlevels, const OrbPointSet* pts, float th,
                                      const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds,
                                      int32_t* best_idx, int32_t* best_dist, OrbProbe* probes_out) {
    if (pts && pts->n > 0 && (!best_idx || !best_dist)) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (kind != ORBHIP_PROBES_FUSE && kind != ORBHIP_PROBES_SIM3)
        return orb_fail(ORBHIP_EINVAL, "the independent search goes with ORBHIP_PROBES_FUSE or ORBHIP_PROBES_SIM3");
    if (pts && pts->n == 0) return ORBHIP_OK;
    ProjectStage st;
    int rc = project_stage(m, kind, cam, cam_to, scale_factors, nlevels, pts, th, 0.f, kp2, desc2, n2, bounds, nullptr, sizeof(int32_t),
                           true, st);
    if (rc) return rc;
    char* b = st.b;
    if ((rc = orbhip_best_in_radius_device(m, (const OrbProbe*)(b + st.o_pr), (const uint8_t*)(b + st.o_pd),
                                           (const int32_t*)(b + st.o_n), st.pcap, (const OrbKeypoint*)(b + st.o_kp),
                                           (const uint8_t*)(b + st.o_d), (const int32_t*)(b + st.o_cs),
                                           (const int32_t*)(b + st.o_ci), 1, st.cap, bounds, (int32_t*)(b + st.o_a),
                                           (int32_t*)(b + st.o_b))))
        return rc;
    HIPCHK(hipMemcpyAsync(best_idx, b + st.o_a, sizeof(int32_t) * pts->n, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(best_dist, b + st.o_b, sizeof(int32_t) * pts->n, hipMemcpyDeviceToHost, m->stream));
    if (probes_out) HIPCHK(hipMemcpyAsync(probes_out, b + st.o_pr, sizeof(OrbProbe) * pts->n, hipMemcpyDeviceToHost, m->stream));
    return orbhip_matcher_synchronize(m);
}

int orbhip_search_by_bow(orbhip_matcher* m, const OrbKeypoint* kp1, const uint8_t* desc1, int n1, const uint8_t* valid1,
                         const int32_t* fv1_nodes, const int32_t* fv1_start, const int32_t* fv1_feat, int fv1_count,
                         const OrbKeypoint* kp2, const uint8_t* desc2, int n2, const uint8_t* valid2, const int32_t* fv2_nodes,
                         const int32_t* fv2_start, const int32_t* fv2_feat, int fv2_count, float nnratio, int check_orientation,
                         int th_low_inclusive, int32_t* matches12, int32_t* matches21, int32_t* nmatches) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (n1 < 0 || n2 < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (!nmatches || (n1 && (!kp1 || !desc1 || !valid1 || !matches12)) || (n2 && (!kp2 || !desc2 || !matches21)))
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    const FvHost fv[2] = {{fv1_nodes, fv1_start, fv1_feat, fv1_count}, {fv2_nodes, fv2_start, fv2_feat, fv2_count}};
    int rc;
    if ((rc = check_fv(fv[0], n1, "first FeatureVector")) || (rc = check_fv(fv[1], n2, "second FeatureVector"))) return rc;
    const int cap = std::max(1, std::max(n1, n2));
    if (cap > ORBHIP_FRAME_MAX_KEYPOINTS) return orb_fail(ORBHIP_ECAP, "%d features > %d per frame", cap, ORBHIP_FRAME_MAX_KEYPOINTS);
    const int node_cap = std::max(1, std::max(fv1_count, fv2_count));
    HIPCHK(hipSetDevice(m->device));
    Arena a;
    size_t o_kp[2], o_d[2], o_v[2], o_nodes[2], o_start[2], o_feat[2];
    for (int s = 0; s < 2; s++) {
        o_kp[s] = a.take(sizeof(OrbKeypoint) * cap);
        o_d[s] = a.take(32 * (size_t)cap);
        o_v[s] = a.take(cap);
        o_nodes[s] = a.take(sizeof(int32_t) * node_cap);
        o_start[s] = a.take(sizeof(int32_t) * (node_cap + 1));
        o_feat[s] = a.take(sizeof(int32_t) * cap);
    }
    const size_t o_n = a.take(sizeof(int32_t) * 4), o_m12 = a.take(sizeof(int32_t) * cap), o_m21 = a.take(sizeof(int32_t) * cap);
    const size_t o_nm = a.take(sizeof(int32_t));
    if ((rc = reserve_stage(m, a.off))) return rc;
    char* b = (char*)m->d_stage;
    const int32_t counts[4] = {n1, n2, fv1_count, fv2_count};
    HIPCHK(hipMemcpyAsync(b + o_n, counts, sizeof(counts), hipMemcpyHostToDevice, m->stream));
    const OrbKeypoint* kp[2] = {kp1, kp2};
    const uint8_t* desc[2] = {desc1, desc2};
    const uint8_t* valid[2] = {valid1, valid2};
    const int n[2] = {n1, n2};
    for (int s = 0; s < 2; s++) {
        if (n[s]) {
            HIPCHK(hipMemcpyAsync(b + o_kp[s], kp[s], sizeof(OrbKeypoint) * n[s], hipMemcpyHostToDevice, m->stream));
            HIPCHK(hipMemcpyAsync(b + o_d[s], desc[s], 32 * (size_t)n[s], hipMemcpyHostToDevice, m->stream));
            if (valid[s]) HIPCHK(hipMemcpyAsync(b + o_v[s], valid[s], n[s], hipMemcpyHostToDevice, m->stream));
        }
        if (fv[s].count) {
            HIPCHK(hipMemcpyAsync(b + o_nodes[s], fv[s].nodes, sizeof(int32_t) * fv[s].count, hipMemcpyHostToDevice, m->stream));
            HIPCHK(hipMemcpyAsync(b + o_start[s], fv[s].start, sizeof(int32_t) * (fv[s].count + 1), hipMemcpyHostToDevice, m->stream));
            if (fv[s].start[fv[s].count])
                HIPCHK(hipMemcpyAsync(b + o_feat[s], fv[s].feat, sizeof(int32_t) * fv[s].start[fv[s].count], hipMemcpyHostToDevice,
                                      m->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(m->stream));  // counts[] lives on this stack frame
    const int32_t* d_n = (const int32_t*)(b + o_n);
    rc = orbhip_search_by_bow_device(
        m, (const OrbKeypoint*)(b + o_kp[0]), (const uint8_t*)(b + o_d[0]), d_n, (const uint8_t*)(b + o_v[0]),
        (const int32_t*)(b + o_nodes[0]), (const int32_t*)(b + o_start[0]), (const int32_t*)(b + o_feat[0]), d_n + 2,
        (const OrbKeypoint*)(b + o_kp[1]), (const uint8_t*)(b + o_d[1]), d_n + 1, valid2 ? (const uint8_t*)(b + o_v[1]) : nullptr,
        (const int32_t*)(b + o_nodes[1]), (const int32_t*)(b + o_start[1]), (const int32_t*)(b + o_feat[1]), d_n + 3, 1, cap,
        node_cap, nnratio, check_orientation, th_low_inclusive, (int32_t*)(b + o_m12), (int32_t*)(b + o_m21), (int32_t*)(b + o_nm));
    if (rc) return rc;
    if (n1) HIPCHK(hipMemcpyAsync(matches12, b + o_m12, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, m->stream));
    if (n2) HIPCHK(hipMemcpyAsync(matches21, b + o_m21, sizeof(int32_t) * n2, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(nmatches, b + o_nm, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    return orbhip_matcher_synchronize(m);
}

int orbhip_search_for_triangulation(orbhip_matcher* m, const OrbKeypoint* kp1, const uint8_t* desc1, int n1, const uint8_t* free1,
                                    const int32_t* fv1_nodes, const int32_t* fv1_start, const int32_t* fv1_feat, int fv1_count,
                                    const OrbKeypoint* kp2, const uint8_t* desc2, int n2, const uint8_t* free2,
                                    const int32_t* fv2_nodes, const int32_t* fv2_start, const int32_t* fv2_feat, int fv2_count,
                                    const float* F12, const float* level_sigma2, int nlevels, int check_orientation,
                                    int32_t* matches12, int32_t* nmatches) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (n1 < 0 || n2 < 0 || nlevels < 1) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (!nmatches || !F12 || !level_sigma2 || (n1 && (!kp1 || !desc1 || !free1 || !matches12)) || (n2 && (!kp2 || !desc2 || !free2)))
        return orb_fail(ORBHIP_EINVAL, "null buffer");
    const FvHost fv[2] = {{fv1_nodes, fv1_start, fv1_feat, fv1_count}, {fv2_nodes, fv2_start, fv2_feat, fv2_count}};
    int rc;
    if ((rc = check_fv(fv[0], n1, "first FeatureVector")) || (rc = check_fv(fv[1], n2, "second FeatureVector"))) return rc;
    for (int i = 0; i < n2; i++)
        if (kp2[i].octave < 0 || kp2[i].octave >= nlevels)
            return orb_fail(ORBHIP_EINVAL, "keypoint %d of the second keyframe: octave %d of %d levels", i, kp2[i].octave, nlevels);
    const int cap = std::max(1, std::max(n1, n2));
    if (cap > ORBHIP_FRAME_MAX_KEYPOINTS) return orb_fail(ORBHIP_ECAP, "%d features > %d per frame", cap, ORBHIP_FRAME_MAX_KEYPOINTS);
    const int node_cap = std::max(1, std::max(fv1_count, fv2_count));
    HIPCHK(hipSetDevice(m->device));
    Arena a;
    size_t o_kp[2], o_d[2], o_v[2], o_nodes[2], o_start[2], o_feat[2];
    for (int s = 0; s < 2; s++) {
        o_kp[s] = a.take(sizeof(OrbKeypoint) * cap);
        o_d[s] = a.take(32 * (size_t)cap);
        o_v[s] = a.take(cap);
        o_nodes[s] = a.take(sizeof(int32_t) * node_cap);
        o_start[s] = a.take(sizeof(int32_t) * (node_cap + 1));
        o_feat[s] = a.take(sizeof(int32_t) * cap);
    }
    const size_t o_n = a.take(sizeof(int32_t) * 4), o_m12 = a.take(sizeof(int32_t) * cap), o_m21 = a.take(sizeof(int32_t) * cap);
    const size_t o_nm = a.take(sizeof(int32_t)), o_F = a.take(sizeof(float) * 9), o_s2 = a.take(sizeof(float) * nlevels);
    if ((rc = reserve_stage(m, a.off))) return rc;
    char* b = (char*)m->d_stage;
    const int32_t counts[4] = {n1, n2, fv1_count, fv2_count};
    HIPCHK(hipMemcpyAsync(b + o_n, counts, sizeof(counts), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(b + o_F, F12, sizeof(float) * 9, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(b + o_s2, level_sigma2, sizeof(float) * nlevels, hipMemcpyHostToDevice, m->stream));
    const OrbKeypoint* kp[2] = {kp1, kp2};
    const uint8_t* desc[2] = {desc1, desc2};
    const uint8_t* fr[2] = {free1, free2};
    const int n[2] = {n1, n2};
    for (int s = 0; s < 2; s++) {
        if (n[s]) {
            HIPCHK(hipMemcpyAsync(b + o_kp[s], kp[s], sizeof(OrbKeypoint) * n[s], hipMemcpyHostToDevice, m->stream));
            HIPCHK(hipMemcpyAsync(b + o_d[s], desc[s], 32 * (size_t)n[s], hipMemcpyHostToDevice, m->stream));
            HIPCHK(hipMemcpyAsync(b + o_v[s], fr[s], n[s], hipMemcpyHostToDevice, m->stream));
        }
        if (fv[s].count) {
            HIPCHK(hipMemcpyAsync(b + o_nodes[s], fv[s].nodes, sizeof(int32_t) * fv[s].count, hipMemcpyHostToDevice, m->stream));
            HIPCHK(hipMemcpyAsync(b + o_start[s], fv[s].start, sizeof(int32_t) * (fv[s].count + 1), hipMemcpyHostToDevice, m->stream));
            if (fv[s].start[fv[s].count])
                HIPCHK(hipMemcpyAsync(b + o_feat[s], fv[s].feat, sizeof(int32_t) * fv[s].start[fv[s].count], hipMemcpyHostToDevice,
                                      m->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(m->stream));  // counts[] lives on this stack frame
    const int32_t* d_n = (const int32_t*)(b + o_n);
    rc = orbhip_search_for_triangulation_device(
        m, (const OrbKeypoint*)(b + o_kp[0]), (const uint8_t*)(b + o_d[0]), d_n, (const uint8_t*)(b + o_v[0]),
        (const int32_t*)(b + o_nodes[0]), (const int32_t*)(b + o_start[0]), (const int32_t*)(b + o_feat[0]), d_n + 2,
        (const OrbKeypoint*)(b + o_kp[1]), (const uint8_t*)(b + o_d[1]), d_n + 1, (const uint8_t*)(b + o_v[1]),
        (const int32_t*)(b + o_nodes[1]), (const int32_t*)(b + o_start[1]), (const int32_t*)(b + o_feat[1]), d_n + 3, 1, cap,
        node_cap, (const float*)(b + o_F), (const float*)(b + o_s2), nlevels, check_orientation, (int32_t*)(b + o_m12),
        (int32_t*)(b + o_m21), (int32_t*)(b + o_nm));
    if (rc) return rc;
    if (n1) HIPCHK(hipMemcpyAsync(matches12, b + o_m12, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(nmatches, b + o_nm, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    return orbhip_matcher_synchronize(m);
}

int orbhip_distinctive_descriptors(orbhip_matcher* m, const uint8_t* desc, const int32_t* start, int npoints, int32_t* best_idx,
                                   int32_t* best_median) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (npoints < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (npoints == 0) return ORBHIP_OK;
    if (!start || !best_idx || !best_median) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (start[0] != 0) return orb_fail(ORBHIP_EINVAL, "start[0] = %d", start[0]);
    for (int p = 0; p < npoints; p++)
        if (start[p + 1] < start[p]) return orb_fail(ORBHIP_EINVAL, "starts decrease at point %d", p);
    const int total = start[npoints];
    if (total && !desc) return orb_fail(ORBHIP_EINVAL, "null descriptors");
    HIPCHK(hipSetDevice(m->device));
    Arena a;
    const size_t o_d = a.take(32 * (size_t)std::max(total, 1)), o_s = a.take(sizeof(int32_t) * (npoints + 1));
    const size_t o_i = a.take(sizeof(int32_t) * npoints), o_med = a.take(sizeof(int32_t) * npoints);
    int rc = reserve_stage(m, a.off);
    if (rc) return rc;
    char* b = (char*)m->d_stage;
    if (total) HIPCHK(hipMemcpyAsync(b + o_d, desc, 32 * (size_t)total, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(b + o_s, start, sizeof(int32_t) * (npoints + 1), hipMemcpyHostToDevice, m->stream));
    if ((rc = orbhip_distinctive_descriptors_device(m, (const uint8_t*)(b + o_d), (const int32_t*)(b + o_s), npoints,
                                                    (int32_t*)(b + o_i), (int32_t*)(b + o_med))))
        return rc;
    HIPCHK(hipMemcpyAsync(best_idx, b + o_i, sizeof(int32_t) * npoints, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(best_median, b + o_med, sizeof(int32_t) * npoints, hipMemcpyDeviceToHost, m->stream));
    return orbhip_matcher_synchronize(m);
}

int orbhip_undistort_keypoints(orbhip_matcher* m, const OrbKeypoint* kp, int n, OrbCamera cam, OrbKeypoint* kp_un) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (n < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (n == 0) return ORBHIP_OK;
    if (!kp || !kp_un) return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    Arena a;
    const size_t o_kp = a.take(sizeof(OrbKeypoint) * n), o_n = a.take(sizeof(int32_t));
    int rc = reserve_stage(m, a.off);
    if (rc) return rc;
    char* b = (char*)m->d_stage;
    const int32_t cnt = n;
    HIPCHK(hipMemcpyAsync(b + o_n, &cnt, sizeof(cnt), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(b + o_kp, kp, sizeof(OrbKeypoint) * n, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));  // cnt lives on this stack frame
    if ((rc = orbhip_undistort_keypoints_device(m, (const OrbKeypoint*)(b + o_kp), (const int32_t*)(b + o_n), 1, n, cam,
                                                (OrbKeypoint*)(b + o_kp))))
        return rc;
    HIPCHK(hipMemcpyAsync(kp_un, b + o_kp, sizeof(OrbKeypoint) * n, hipMemcpyDeviceToHost, m->stream));
    return orbhip_matcher_synchronize(m);
}

int orbhip_bow_transform(orbhip_matcher* m, const orbhip_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word,
                         int32_t* node, double* weight) {
    if (!m || !v) return orb_fail(ORBHIP_EINVAL, "null handle");
    if (n < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (n == 0) return ORBHIP_OK;
    if (!desc || !word || !node || !weight) return orb_fail(ORBHIP_EINVAL, "null buffer");
    HIPCHK(hipSetDevice(m->device));
    Arena a;
    const size_t o_d = a.take(32 * (size_t)n), o_w = a.take(sizeof(int32_t) * n), o_nd = a.take(sizeof(int32_t) * n);
    const size_t o_wt = a.take(sizeof(double) * n);
    int rc = reserve_stage(m, a.off);
    if (rc) return rc;
    char* b = (char*)m->d_stage;
    HIPCHK(hipMemcpyAsync(b + o_d, desc, 32 * (size_t)n, hipMemcpyHostToDevice, m->stream));
    if ((rc = orbhip_bow_transform_device(m, v, (const uint8_t*)(b + o_d), n, levelsup, (int32_t*)(b + o_w), (int32_t*)(b + o_nd),
                                          (double*)(b + o_wt))))
        return rc;
    HIPCHK(hipMemcpyAsync(word, b + o_w, sizeof(int32_t) * n, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(node, b + o_nd, sizeof(int32_t) * n, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipMemcpyAsync(weight, b + o_wt, sizeof(double) * n, hipMemcpyDeviceToHost, m->stream));
    return orbhip_matcher_synchronize(m);
}

}  // extern "C"
