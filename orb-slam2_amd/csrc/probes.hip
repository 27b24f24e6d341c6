// The lines of the reference's projection matchers that turn a map point into a search window, for gfx950: one
// lane per map point evaluates what the reference evaluates between its pointer tests and its GetFeaturesInArea
// call and writes the OrbProbe that orbhip_search_by_projection_device / orbhip_best_in_radius_device take.
//   ORBHIP_PROBES_LAST_FRAME  ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th)      src/ORBmatcher.cc:1531-1556
//   ORBHIP_PROBES_KEYFRAME    ORBmatcher::SearchByProjection(CurrentFrame, pKF, found, th, ..)  :1651-1690
//   ORBHIP_PROBES_FRUSTUM     Frame::isInFrustum (src/Frame.cc:138-199) followed by
//                             ORBmatcher::SearchByProjection(F, vpMapPoints, th)                :63-81
//   ORBHIP_PROBES_FUSE        ORBmatcher::Fuse, both forms (:1054-1094, :1180-1223), and
//                             ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (:326-375)
//   ORBHIP_PROBES_WINDOW      ORBmatcher::SearchByProjection(F1, F2, windowSize, vpMapPointMatches2) (:545-566)
//   ORBHIP_PROBES_SIM3        ORBmatcher::SearchBySim3, either direction (:1330-1375, :1415-1460): the point goes into
//                             the camera of its own keyframe first and from there through the similarity
// cv::Mat expressions are evaluated as OpenCV 2.4 evaluates them for these shapes (CV_32F): R*x + t through the
// small-matrix path of cv::gemm (each row a float sum left to right, then (float)(sum*1.0 + t*1.0) in double);
// cv::norm and Mat::dot with double accumulation; "1.0/z" in double stored to float.  The file is compiled without
// contraction, so every expression below rounds where the reference's scalar code rounds.
#include <hip/hip_runtime.h>

#include <cmath>

#include "matcher_internal.h"

namespace {

struct KpDevP {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

__host__ __device__ inline void rigid(const OrbProjection& c, const float* x, float out[3]) {
    for (int i = 0; i < 3; i++) {
        const float* a = c.Rcw + 3 * i;
        const float t0 = a[0] * x[0] + a[1] * x[1] + a[2] * x[2];
        out[i] = (float)(t0 * 1.0 + c.tcw[i] * 1.0);
    }
}

__host__ __device__ inline double norm3(const float* v) {
    double s0 = 0;
    for (int i = 0; i < 3; i++) s0 += (double)v[i] * v[i];
    return sqrt(s0);
}

__host__ __device__ inline double dot3(const float* a, const float* b) {
    double r = 0;
    for (int i = 0; i < 3; i++) r += (double)a[i] * b[i];
    return r;
}

// std::lower_bound(vfScaleFactors.begin(), vfScaleFactors.end(), ratio) - begin, clamped to the last level
__device__ inline int predicted_level(const float* __restrict__ sf, int nlevels, float ratio) {
    int lo = 0, hi = nlevels;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sf[mid] < ratio) lo = mid + 1;
        else hi = mid;
    }
    return min(lo, nlevels - 1);
}

__device__ inline void clear_probe(OrbProbe& q) {
    q.u = q.v = q.r = q.angle = 0.f;
    q.min_level = q.max_level = q.valid = q.reserved = 0;
}

__global__ __launch_bounds__(256) void k_build_probes(int kind, const OrbProjection* __restrict__ cams,
                                                      const OrbProjection* __restrict__ cams_to, OrbFrameBounds bounds,
                                                      const float* __restrict__ sf, int nlevels, const float* __restrict__ pos,
                                                      const float* __restrict__ normal, const float* __restrict__ min_dist,
                                                      const float* __restrict__ max_dist, const KpDevP* __restrict__ src_kp,
                                                      const uint8_t* __restrict__ skip, const int32_t* __restrict__ npoints,
                                                      int pcap, float th, float viewing_cos_limit, OrbProbe* __restrict__ probes) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= pcap) return;
    const long long at = (long long)f * pcap + i;
    OrbProbe q;
    clear_probe(q);
    // every early exit below is a `continue` / `return false` of the reference: the probe stays invalid
    do {
        if (i >= npoints[f] || (skip && skip[at])) break;
        OrbProjection c = cams[f];
        const float* P = pos + at * 3;
        float Pc[3];
        rigid(c, P, Pc);
        if (kind == ORBHIP_PROBES_SIM3) {
            // p3Dc1 = R1w*p3Dw + t1w, then p3Dc2 = sR21*p3Dc1 + t21: two products, each rounded to float
            const float Pf[3] = {Pc[0], Pc[1], Pc[2]};
            c = cams_to[f];
            rigid(c, Pf, Pc);
        }
        float u, v;
        if (kind == ORBHIP_PROBES_WINDOW) {
            // no depth test, no image test: the window is centred wherever the point lands
            const float invz = 1.0 / Pc[2];
            u = c.fx * Pc[0] * invz + c.cx;
            v = c.fy * Pc[1] * invz + c.cy;
        } else if (kind == ORBHIP_PROBES_FUSE || kind == ORBHIP_PROBES_SIM3) {
            if (Pc[2] < 0.0f) break;                    // "Depth must be positive"
            const float invz = 1 / Pc[2];
            const float x = Pc[0] * invz;
            const float y = Pc[1] * invz;
            u = c.fx * x + c.cx;
            v = c.fy * y + c.cy;
            if (!(u >= bounds.min_x && u < bounds.max_x && v >= bounds.min_y && v < bounds.max_y)) break;  // KeyFrame::IsInImage
        } else {
            if (kind == ORBHIP_PROBES_FRUSTUM && Pc[2] < 0.0) break;  // "Check positive depth"
            const float invz = 1.0 / Pc[2];
            u = c.fx * Pc[0] * invz + c.cx;
            v = c.fy * Pc[1] * invz + c.cy;
            if (u < bounds.min_x || u > bounds.max_x) break;
            if (v < bounds.min_y || v > bounds.max_y) break;
        }
        int level_lo, level_hi, level;
        float angle = 0.f;
        if (kind == ORBHIP_PROBES_WINDOW) {
            level = level_lo = level_hi = src_kp[at].octave;  // F1.mvKeysUn[i1].octave, the only level searched
            q.r = th;                                        // windowSize
        } else if (kind == ORBHIP_PROBES_SIM3) {
            const float dist = norm3(Pc);  // distance in the target camera (the similarity scales it)
            const float minDistance = min_dist[at], maxDistance = max_dist[at];
            if (dist < minDistance || dist > maxDistance) break;
            const float ratio = dist / minDistance;
            level = predicted_level(sf, nlevels, ratio);
            level_lo = level - 1;
            level_hi = level;
            q.r = th * sf[level];
        } else if (kind == ORBHIP_PROBES_LAST_FRAME) {
            const int nPredictedOctave = min(max(src_kp[at].octave, 0), nlevels - 1);  // LastFrame.mvKeys[i].octave
            level = nPredictedOctave;
            level_lo = nPredictedOctave - 1;
            level_hi = nPredictedOctave + 1;
            angle = src_kp[at].angle;
            q.r = th * sf[level];
        } else {
            const float PO[3] = {P[0] - c.Ow[0], P[1] - c.Ow[1], P[2] - c.Ow[2]};
            const float dist = norm3(PO);
            const float minDistance = min_dist[at];
            float viewCos = 0.f;
            if (kind != ORBHIP_PROBES_KEYFRAME) {
                const float maxDistance = max_dist[at];
                if (dist < minDistance || dist > maxDistance) break;  // scale invariance region of the map point
                if (kind == ORBHIP_PROBES_FRUSTUM) {
                    viewCos = dot3(PO, normal + at * 3) / dist;
                    if (viewCos < viewing_cos_limit) break;
                } else if (dot3(PO, normal + at * 3) < 0.5 * dist) {
                    break;  // "Viewing angle must be less than 60 deg"
                }
            }
            const float ratio = dist / minDistance;
            level = predicted_level(sf, nlevels, ratio);
            level_lo = level - 1;
            if (kind == ORBHIP_PROBES_KEYFRAME) {
                level_hi = level + 1;
                angle = src_kp[at].angle;  // pKF->GetKeyPointUn(i).angle
                q.r = th * sf[level];
            } else if (kind == ORBHIP_PROBES_FRUSTUM) {
                level_hi = level;
                float r = viewCos > 0.998 ? 2.5 : 4.0;  // ORBmatcher::RadiusByViewingCos, src/ORBmatcher.cc:130-136
                if (th != 1.0) r *= th;
                q.r = r * sf[level];
            } else {
                level_hi = level;
                q.r = th * sf[level];
            }
        }
        q.u = u;
        q.v = v;
        q.min_level = level_lo;
        q.max_level = level_hi;
        q.angle = angle;
        q.valid = 1;
    } while (false);
    probes[at] = q;
}

}  // namespace

extern "C" {

int orbhip_projection_from_pose(const float* Tcw, float fx, float fy, float cx, float cy, OrbProjection* out) {
    if (!Tcw || !out) return orb_fail(ORBHIP_EINVAL, "null argument");
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) out->Rcw[3 * i + j] = Tcw[4 * i + j];
        out->tcw[i] = Tcw[4 * i + 3];
    }
    // Ow = -Rcw.t()*tcw: cv::gemm with a transposed first factor and alpha = -1 (general path: double accumulation,
    // times alpha, one rounding)
    for (int i = 0; i < 3; i++) {
        double s0 = 0;
        for (int k = 0; k < 3; k++) s0 += (double)out->Rcw[3 * k + i] * (double)out->tcw[k];
        out->Ow[i] = (float)(s0 * -1.0);
    }
    out->fx = fx;
    out->fy = fy;
    out->cx = cx;
    out->cy = cy;
    return ORBHIP_OK;
}

int orbhip_projection_from_sim3(const float* Scw, float fx, float fy, float cx, float cy, OrbProjection* out) {
    if (!Scw || !out) return orb_fail(ORBHIP_EINVAL, "null argument");
    // "Decompose Scw" (src/ORBmatcher.cc:305-310, :1156-1160): scw = sqrt(sRcw.row(0).dot(sRcw.row(0))) with the dot
    // product in double; M / scw is an elementwise float multiply by (float)(1.0 / scw)
    double d = 0;
    for (int j = 0; j < 3; j++) d += (double)Scw[j] * Scw[j];
    const float scw = sqrt(d);
    const float inv = (float)(1. / scw);
    float T[16] = {0};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[4 * i + j] = Scw[4 * i + j] * inv;
        T[4 * i + 3] = Scw[4 * i + 3] * inv;
    }
    return orbhip_projection_from_pose(T, fx, fy, cx, cy, out);
}

// Checks shared by the two device forms; cams_to only for ORBHIP_PROBES_SIM3.
static int build_probes_launch(orbhip_matcher* m, int kind, const OrbProjection* d_cams, const OrbProjection* d_cams_to,
                               OrbFrameBounds bounds, const float* d_scale_factors, int nlevels, const float* d_pos,
                               const float* d_normal, const float* d_min_dist, const float* d_max_dist,
                               const OrbKeypoint* d_src_kp, const uint8_t* d_skip, const int32_t* d_npoints, int batch,
                               int probe_cap, float th, float viewing_cos_limit, OrbProbe* d_probes) {
    if (!m) return orb_fail(ORBHIP_EINVAL, "null matcher");
    if (kind < ORBHIP_PROBES_LAST_FRAME || kind > ORBHIP_PROBES_SIM3) return orb_fail(ORBHIP_EINVAL, "unknown probe kind %d", kind);
    if (batch < 0 || probe_cap < 1 || nlevels < 1) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (batch == 0) return ORBHIP_OK;
    if (!d_cams || !d_scale_factors || !d_pos || !d_npoints || !d_probes) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if ((kind == ORBHIP_PROBES_SIM3) != (d_cams_to != nullptr))
        return orb_fail(ORBHIP_EINVAL, "a second camera goes with ORBHIP_PROBES_SIM3 and with no other kind");
    if ((kind == ORBHIP_PROBES_LAST_FRAME || kind == ORBHIP_PROBES_KEYFRAME || kind == ORBHIP_PROBES_WINDOW) && !d_src_kp)
        return orb_fail(ORBHIP_EINVAL, "this kind reads the octave / angle of a source keypoint per point");
    if (kind != ORBHIP_PROBES_LAST_FRAME && kind != ORBHIP_PROBES_WINDOW && !d_min_dist)
        return orb_fail(ORBHIP_EINVAL, "null distance range");
    if ((kind == ORBHIP_PROBES_FRUSTUM || kind == ORBHIP_PROBES_FUSE) && (!d_max_dist || !d_normal))
        return orb_fail(ORBHIP_EINVAL, "this kind reads the normal and the distance range of every point");
    if (kind == ORBHIP_PROBES_SIM3 && !d_max_dist) return orb_fail(ORBHIP_EINVAL, "null distance range");
    HIPCHK(hipSetDevice(m->device));
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_build_probes, dim3((probe_cap + 255) / 256, batch), dim3(256), 0, m->stream, kind, d_cams, d_cams_to,
                       bounds, d_scale_factors, nlevels, d_pos, d_normal, d_min_dist, d_max_dist, (const KpDevP*)d_src_kp, d_skip,
                       d_npoints, probe_cap, th, viewing_cos_limit, d_probes);
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

int orbhip_build_probes_device(orbhip_matcher* m, int kind, const OrbProjection* d_cams, OrbFrameBounds bounds,
                               const float* d_scale_factors, int nlevels, const float* d_pos, const float* d_normal,
                               const float* d_min_dist, const float* d_max_dist, const OrbKeypoint* d_src_kp,
                               const uint8_t* d_skip, const int32_t* d_npoints, int batch, int probe_cap, float th,
                               float viewing_cos_limit, OrbProbe* d_probes) {
    if (kind == ORBHIP_PROBES_SIM3) return orb_fail(ORBHIP_EINVAL, "ORBHIP_PROBES_SIM3 takes two cameras: orbhip_build_probes_sim3_device");
    return build_probes_launch(m, kind, d_cams, nullptr, bounds, d_scale_factors, nlevels, d_pos, d_normal, d_min_dist, d_max_dist,
                               d_src_kp, d_skip, d_npoints, batch, probe_cap, th, viewing_cos_limit, d_probes);
}

int orbhip_build_probes_sim3_device(orbhip_matcher* m, const OrbProjection* d_cams_from, const OrbProjection* d_cams_to,
                                    OrbFrameBounds bounds, const float* d_scale_factors, int nlevels, const float* d_pos,
                                    const float* d_min_dist, const float* d_max_dist, const uint8_t* d_skip,
                                    const int32_t* d_npoints, int batch, int probe_cap, float th, OrbProbe* d_probes) {
    if (!d_cams_to) return orb_fail(ORBHIP_EINVAL, "null buffer");
    return build_probes_launch(m, ORBHIP_PROBES_SIM3, d_cams_from, d_cams_to, bounds, d_scale_factors, nlevels, d_pos, nullptr,
                               d_min_dist, d_max_dist, nullptr, d_skip, d_npoints, batch, probe_cap, th, 0.f, d_probes);
}

int orbhip_projection_sim3_pair(float s12, const float* R12, const float* t12, float fx, float fy, float cx, float cy,
                                OrbProjection* to2, OrbProjection* to1) {
    if (!R12 || !t12 || !to2 || !to1) return orb_fail(ORBHIP_EINVAL, "null argument");
    // sR12 = s12*R12 and sR21 = (1.0/s12)*R12.t() (src/ORBmatcher.cc:1290-1291): elementwise float products with the
    // scale converted to float; t21 = -sR21*t12 (:1292): one small-matrix product, float accumulation, times -1.0
    const float a12 = (float)(double)s12, a21 = (float)(1.0 / s12);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            to1->Rcw[3 * i + j] = R12[3 * i + j] * a12;
            to2->Rcw[3 * i + j] = R12[3 * j + i] * a21;
        }
    for (int i = 0; i < 3; i++) {
        const float* a = to2->Rcw + 3 * i;
        const float t0 = a[0] * t12[0] + a[1] * t12[1] + a[2] * t12[2];
        to2->tcw[i] = (float)(t0 * -1.0);
        to1->tcw[i] = t12[i];
        to1->Ow[i] = to2->Ow[i] = 0.f;  // not read by ORBHIP_PROBES_SIM3
    }
    to1->fx = to2->fx = fx;
    to1->fy = to2->fy = fy;
    to1->cx = to2->cx = cx;
    to1->cy = to2->cy = cy;
    return ORBHIP_OK;
}

int orbhip_probes_from_tracked(int n, const float* proj_xy, const float* view_cos, const int32_t* level, const uint8_t* skip,
                               float th, const float* scale_factors, int nlevels, OrbProbe* probes) {
    if (n < 0 || nlevels < 1) return orb_fail(ORBHIP_EINVAL, "bad size");
    if (n == 0) return ORBHIP_OK;
    if (!proj_xy || !view_cos || !level || !scale_factors || !probes) return orb_fail(ORBHIP_EINVAL, "null buffer");
    for (int i = 0; i < n; i++) {
        OrbProbe q = {0.f, 0.f, 0.f, 0, 0, 0.f, 0, 0};
        if (!(skip && skip[i])) {
            const int l = level[i];
            if (l < 0 || l >= nlevels) return orb_fail(ORBHIP_EINVAL, "point %d: predicted level %d of %d", i, l, nlevels);
            float r = view_cos[i] > 0.998 ? 2.5 : 4.0;  // ORBmatcher::RadiusByViewingCos, src/ORBmatcher.cc:130-136
            if (th != 1.0) r *= th;                     // bFactor, :54 and :77-78
            q.u = proj_xy[2 * i];
            q.v = proj_xy[2 * i + 1];
            q.r = r * scale_factors[l];
            q.min_level = l - 1;
            q.max_level = l;
            q.valid = 1;
        }
        probes[i] = q;
    }
    return ORBHIP_OK;
}

}  // extern "C"
