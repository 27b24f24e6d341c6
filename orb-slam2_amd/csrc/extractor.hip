// Host side of the extractor: constructor tables, level/cell geometry, device arenas and the
// kernel sequence of one chunk of frames.  Implements the orbhip_extractor_* / orbhip_extract*
// entry points of include/orbhip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbhip.h"
#include "common.h"
#include "extract_kernels.h"

#define ORB_NSTAGES 8  // level0, resize, score, nms, select, orient, blur, describe

namespace {

const signed char kPattern[1024] = {
#include "../../include/orb_pattern.inc"
};

inline int round_half_even(double v) { return (int)std::nearbyint(v); }  // cvRound
inline int floor_int(double v) { int i = (int)v; return i - (i > v); }
inline int ceil_int(double v) { int i = (int)v; return i + (i < v); }
// inverse of the order-preserving float -> unsigned map of the HARRIS_SCORE keys: IEEE bits of the response
inline uint32_t ordered_bits_to_float_bits(uint32_t u) { return (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; }

// Bumped whenever a buffer of any handle is (re)allocated or released: a captured launch sequence holds raw
// addresses and is only replayed while this is what it was at capture time.
static std::atomic<unsigned long long> g_buffer_generation{0};

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return ORBHIP_OK;
        g_buffer_generation++;
        if (p) hipFree(p);
        p = nullptr;
        bytes = 0;
        if (hipMalloc(&p, need) != hipSuccess) return orb_fail(ORBHIP_ENOMEM, "hipMalloc of %zu bytes failed", need);
        bytes = need;
        return ORBHIP_OK;
    }
    void release() {
        g_buffer_generation++;
        if (p) hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

// Page-locked host staging of the host-pointer entry points (grown on demand, kept in the handle).
struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return ORBHIP_OK;
        g_buffer_generation++;
        if (p) hipHostFree(p);
        p = nullptr;
        bytes = 0;
        if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess)
            return orb_fail(ORBHIP_ENOMEM, "hipHostMalloc of %zu bytes failed", need);
        bytes = need;
        return ORBHIP_OK;
    }
    void release() {
        if (p) hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

}  // namespace

struct orbhip_extractor {
    // constructor state (src/ORBextractor.cc:457-511)
    int nfeatures;
    double scaleFactor;
    int nlevels, scoreType, fastTh;
    std::vector<int> featuresPerLevel;
    std::vector<float> scale, invScale;
    int umax[16];
    int gtaps[7];

    int device;
    hipStream_t stream = nullptr;
    // worker streams: a chunk is split into sub-batches whose kernel sequences run concurrently, so
    // the compute-bound FAST kernel of one sub-batch overlaps the memory-bound stages of another
    std::vector<hipStream_t> workers;
    std::vector<hipEvent_t> worker_done;
    // The blur of a chunk runs beside its selection kernels on a stream of its own: entry 0 belongs to `stream`, entry
    // 1 + i to workers[i].  blur_fork is recorded behind the FAST kernel, blur_join behind the blur.
    std::vector<hipStream_t> blur_streams;
    std::vector<hipEvent_t> blur_fork, blur_join;
    bool blur_beside_selection = true;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool in_capture = false;  // the stream is being captured into a graph: the timing events are recorded around the replay instead
    // host-buffer batches of 512 frames and more go to the device in pieces on a copy stream of their own
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> copy_done;
    float last_ms = 0.f;
    int max_batch;
    // per-stage timing (bench instrumentation): events bracket each kernel group of a chunk
    bool profiling = false;
    bool gauss_tie_even = false;  // ORBHIP_GAUSS_SSE2: ties of the Gaussian column pass round to even
    std::vector<hipEvent_t> prof_ev;           // (ORB_NSTAGES + 1) events per chunk of the last call
    int prof_chunks = 0;
    float stage_ms[ORB_NSTAGES] = {0};
    int stage_launches[ORB_NSTAGES] = {0};

    // geometry of the current image size
    int cur_rows = 0, cur_cols = 0;
    GeomDev hg;
    std::vector<CellDesc> hcells;
    int max_cell_h = 0, max_cell_w = 0;
    bool fused_fast = false;  // every cell fits the LDS-resident fused FAST kernel
    int last_batch = 0;  // frames of the last chunk (for the stage taps)

    DeviceBuf d_geom, d_cells, d_xofs, d_xab, d_ysrc, d_yab, d_disc, d_pattern;
    DeviceBuf d_fcrecs;  // per-cell records of the fused FAST kernel
    DeviceBuf d_pyr, d_blur, d_score, d_cand, d_cand64, d_ntotal, d_nretain, d_lvl, d_lvlcount, d_angles, d_slots, d_err;
    // staging of the host-pointer entry points: frames in, one output block (counts, keypoints, descriptors)
    // on the device and page-locked mirrors of both, so that a call is two copies and one synchronisation
    DeviceBuf d_img, d_outblk;
    PinnedBuf h_img, h_outblk;

    // One frame per call (how src/Frame.cc:61 uses the class): the whole sequence of a call — image copy in, ~20
    // kernels, result block out — captured once per image size as a graph and replayed with one launch.
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey {
        int rows = 0, cols = 0, cap = 0;
        bool tie_even = false;
        unsigned long long generation = 0;
        bool operator==(const GraphKey& o) const {
            return rows == o.rows && cols == o.cols && cap == o.cap && tie_even == o.tie_even && generation == o.generation;
        }
    } graph_key;
    bool graph_off = false;  // capture failed once, or ORBHIP_GRAPH=0: plain launches from then on
};

namespace {

// ORBextractor::ORBextractor: scale tables, per-level quotas, umax.
int build_tables(orbhip_extractor* e) {
    const int nlevels = e->nlevels;
    e->scale.assign(nlevels, 1.f);
    for (int i = 1; i < nlevels; i++) e->scale[i] = (float)(e->scale[i - 1] * e->scaleFactor);
    const float inv = (float)(1.0f / e->scaleFactor);
    e->invScale.assign(nlevels, 1.f);
    for (int i = 1; i < nlevels; i++) e->invScale[i] = e->invScale[i - 1] * inv;

    e->featuresPerLevel.assign(nlevels, 0);
    const float factor = (float)(1.0 / e->scaleFactor);
    float desired = e->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        e->featuresPerLevel[l] = round_half_even(desired);
        sum += e->featuresPerLevel[l];
        desired *= factor;
    }
    e->featuresPerLevel[nlevels - 1] = std::max(e->nfeatures - sum, 0);

    // rows of the radius-15 disc (src/ORBextractor.cc:495-510)
    const int hp = ORB_HALF_PATCH;
    const int vmax = floor_int(hp * std::sqrt(2.f) / 2 + 1);
    const int vmin = ceil_int(hp * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) e->umax[v] = round_half_even(std::sqrt((double)hp * hp - v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    // 8-bit fixed-point taps of getGaussianKernel(7, 2): exp(-x^2/8) in double, stored as float,
    // normalised by the float sum, times 256, rounded half to even.
    float cf[7];
    double s = 0;
    for (int i = 0; i < 7; i++) {
        const double x = i - 3.0;
        cf[i] = (float)std::exp(-0.5 / 4.0 * x * x);
        s += cf[i];
    }
    s = 1. / s;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * s);
        e->gtaps[i] = round_half_even((double)cf[i] * 256);
    }
    return ORBHIP_OK;
}

// Level sizes (src/ORBextractor.cc:786), cell grid (:527-547,560-599), arena layout, resize tables.
int build_geometry(orbhip_extractor* e, int rows, int cols, GeomDev& g, std::vector<CellDesc>& cells,
                   std::vector<int>& xofs, std::vector<uint32_t>& xab, std::vector<uint32_t>& ysrc,
                   std::vector<uint32_t>& yab, int& max_cell_h) {
    memset(&g, 0, sizeof(g));
    cells.clear();
    xofs.clear();
    xab.clear();
    ysrc.clear();
    yab.clear();
    g.nlevels = e->nlevels;
    g.fast_th = e->fastTh;
    if (cols > ORB_MAX_DIM || rows > ORB_MAX_DIM) return orb_fail(ORBHIP_EUNSUPPORTED, "image larger than %d", ORB_MAX_DIM);
    long long plane_off = 0, cand_off = 0, lvl_off = 0;
    int slot = 0;
    max_cell_h = 1;
    const float imageRatio = (float)cols / rows;  // cols, rows of level 0
    for (int l = 0; l < e->nlevels; l++) {
        LevelGeom& L = g.lv[l];
        const float s = e->invScale[l];
        L.w = round_half_even((float)cols * s);
        L.h = round_half_even((float)rows * s);
        if (L.w < 2 * ORB_EDGE + 1 || L.h < 2 * ORB_EDGE + 1)
            return orb_fail(ORBHIP_EGEOM, "level %d is %dx%d: no interior inside the 16-px border", l, L.w, L.h);
        L.stride = (L.w + 2 * ORB_EDGE + 63) & ~63;
        L.ph = L.h + 2 * ORB_EDGE;
        L.plane_off = plane_off;
        plane_off += ((long long)L.stride * L.ph + 255) & ~255LL;
        L.quota = e->featuresPerLevel[l];
        L.scale = e->scale[l];
        L.kp_size = (float)(int)(31 * e->scale[l]);
        L.slot_base = slot;
        slot += L.quota;

        L.cols = (int)std::sqrt((float)L.quota / (5 * imageRatio));
        L.rows = (int)(imageRatio * L.cols);
        if (L.cols < 1 || L.rows < 1)
            return orb_fail(ORBHIP_EGEOM, "level %d: %d features give a %dx%d cell grid", l, L.quota, L.cols, L.rows);
        const int minB = ORB_EDGE, maxBX = L.w - ORB_EDGE, maxBY = L.h - ORB_EDGE;
        const int W = maxBX - minB, H = maxBY - minB;
        L.cellW = (int)std::ceil((float)W / L.cols);
        L.cellH = (int)std::ceil((float)H / L.rows);
        L.nCells = L.rows * L.cols;
        if (L.nCells > ORB_MAX_CELLS_PER_LEVEL) return orb_fail(ORBHIP_EUNSUPPORTED, "level %d has %d cells", l, L.nCells);
        L.nfeaturesCell = (int)std::ceil((float)L.quota / L.nCells);
        L.cell_base = (int)cells.size();
        long long level_cand = 0;
        for (int i = 0; i < L.rows; i++)
            for (int j = 0; j < L.cols; j++) {
                CellDesc c;
                memset(&c, 0, sizeof(c));
                c.level = l;
                c.index = i * L.cols + j;
                c.ini_x = minB + j * L.cellW - 3;
                c.ini_y = minB + i * L.cellH - 3;
                c.x0 = c.ini_x + 3;
                c.y0 = c.ini_y + 3;
                // examined pixels: columns 3 .. hX-4 of the cell image, hX = cellW+6, or up to the
                // border for the last column (hX = maxBorderX + 3 - iniX); same for rows.
                c.cw = (j == L.cols - 1) ? maxBX - c.x0 : L.cellW;
                c.ch = (i == L.rows - 1) ? maxBY - c.y0 : L.cellH;
                // the reference skips a last row/column whose cell image is empty (hX <= 0)
                const int hX = (j == L.cols - 1) ? maxBX + 3 - c.ini_x : L.cellW + 6;
                const int hY = (i == L.rows - 1) ? maxBY + 3 - c.ini_y : L.cellH + 6;
                if (hX <= 0 || hY <= 0) {
                    c.cw = c.ch = 0;
                    c.skipped = 1;
                } else {
                    // a cell image reaching past the level image makes cv::Mat::colRange throw
                    if (c.ini_x + hX > L.w || c.ini_y + hY > L.h)
                        return orb_fail(ORBHIP_EGEOM, "level %d: cell grid does not fit the %dx%d level", l, L.w, L.h);
                    if (c.cw < 0) c.cw = 0;  // 0 < hX <= 6: a visited cell whose image has no examined pixel
                    if (c.ch < 0) c.ch = 0;
                    // ceil() cell sizes can carry a cell that is not in the last row/column past the
                    // border; a corner found there has its orientation patch outside the image, which
                    // the reference reads without a check
                    if (c.cw > 0 && c.ch > 0 && (c.x0 + c.cw > maxBX || c.y0 + c.ch > maxBY))
                        return orb_fail(ORBHIP_EGEOM, "level %d: cell (%d,%d) reaches past the %d-px border of the %dx%d level", l,
                                        i, j, ORB_EDGE, L.w, L.h);
                }
                c.cand_off = cand_off;
                c.cand_cap = ((std::max(c.cw, 0) + 1) / 2) * ((std::max(c.ch, 0) + 1) / 2);
                cand_off += c.cand_cap;
                level_cand += c.cand_cap;
                max_cell_h = std::max(max_cell_h, c.ch);
                cells.push_back(c);
            }
        const long long quota_bound = (long long)L.nfeaturesCell * L.nCells + (long long)L.nCells * (L.nCells + 1) / 2;
        L.lvl_cap = (int)std::min<long long>(level_cand, quota_bound);
        L.lvl_off = lvl_off;
        lvl_off += L.lvl_cap;

        // cv::resize INTER_LINEAR coefficient tables for level l-1 -> l
        L.rz_x_off = (int)xofs.size();
        L.rz_y_off = (int)ysrc.size();
        if (l > 0) {
            const LevelGeom& P = g.lv[l - 1];
            const double scale_x = 1. / ((double)L.w / P.w), scale_y = 1. / ((double)L.h / P.h);
            for (int dx = 0; dx < L.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = floor_int(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= P.w - 1) { fx = 0; sx = P.w - 1; }
                const int a0 = std::min(std::max(round_half_even((1.f - fx) * 2048), -32768), 32767);
                const int a1 = std::min(std::max(round_half_even(fx * 2048), -32768), 32767);
                xofs.push_back(sx);
                xab.push_back((uint32_t)(a0 & 0xffff) | ((uint32_t)(a1 & 0xffff) << 16));
            }
            for (int dy = 0; dy < L.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                const int sy = floor_int(fy);
                fy -= sy;
                const int sy0 = std::min(std::max(sy, 0), P.h - 1), sy1 = std::min(std::max(sy + 1, 0), P.h - 1);
                const int b0 = std::min(std::max(round_half_even((1.f - fy) * 2048), -32768), 32767);
                const int b1 = std::min(std::max(round_half_even(fy * 2048), -32768), 32767);
                ysrc.push_back((uint32_t)sy0 | ((uint32_t)sy1 << 16));
                yab.push_back((uint32_t)(b0 & 0xffff) | ((uint32_t)(b1 & 0xffff) << 16));
            }
        }
    }
    g.nfeatures = slot;
    g.cells_per_frame = (int)cells.size();
    g.frame_bytes = plane_off;
    g.cand_per_frame = cand_off;
    g.lvl_per_frame = lvl_off;
    return ORBHIP_OK;
}

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t err__ = (call);                                                           \
        if (err__ != hipSuccess)                                                             \
            return orb_fail(ORBHIP_EHIP, "%s failed: %s", #call, hipGetErrorString(err__)); \
    } while (0)

int upload(DeviceBuf& b, const void* src, size_t bytes, hipStream_t st) {
    int rc = b.ensure(std::max<size_t>(bytes, 16));
    if (rc) return rc;
    if (bytes) {
        HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return ORBHIP_OK;
}

// (Re)builds geometry for an image size and makes the arenas large enough for `batch` frames.
int set_size(orbhip_extractor* e, int rows, int cols, int batch) {
    if (rows != e->cur_rows || cols != e->cur_cols) {
        GeomDev g;
        std::vector<CellDesc> cells;
        std::vector<int> xofs;
        std::vector<uint32_t> xab, ysrc, yab;
        int mch = 0;
        int rc = build_geometry(e, rows, cols, g, cells, xofs, xab, ysrc, yab, mch);
        if (rc) return rc;
        // a rebuild that fails half way must not leave the handle claiming the old size: force a full rebuild next time
        e->cur_rows = e->cur_cols = 0;
        if ((rc = upload(e->d_geom, &g, sizeof(g), e->stream))) return rc;
        if ((rc = upload(e->d_cells, cells.data(), cells.size() * sizeof(CellDesc), e->stream))) return rc;
        {
            std::vector<unsigned char> recs(cells.size() * orbk::fast_cell_rec_bytes());
            orbk::build_fast_cell_recs(g, cells.data(), (int)cells.size(), recs.data());
            if ((rc = upload(e->d_fcrecs, recs.data(), recs.size(), e->stream))) return rc;
        }
        if ((rc = upload(e->d_xofs, xofs.data(), xofs.size() * sizeof(int), e->stream))) return rc;
        if ((rc = upload(e->d_xab, xab.data(), xab.size() * sizeof(uint32_t), e->stream))) return rc;
        if ((rc = upload(e->d_ysrc, ysrc.data(), ysrc.size() * sizeof(uint32_t), e->stream))) return rc;
        if ((rc = upload(e->d_yab, yab.data(), yab.size() * sizeof(uint32_t), e->stream))) return rc;
        e->hg = g;
        e->hcells.swap(cells);
        e->max_cell_h = mch;
        e->max_cell_w = 0;
        for (const CellDesc& c : e->hcells) e->max_cell_w = std::max(e->max_cell_w, c.cw);
        const char* nf = getenv("ORBHIP_NO_FUSED_FAST");
        e->fused_fast = orbk::fast_cell_supported(e->max_cell_w, e->max_cell_h) && !(nf && nf[0] == '1');
        e->cur_rows = rows;
        e->cur_cols = cols;
    }
    const GeomDev& g = e->hg;
    const size_t B = (size_t)batch;
    int rc;
    if ((rc = e->d_pyr.ensure(B * g.frame_bytes))) return rc;
    if ((rc = e->d_blur.ensure(B * g.frame_bytes))) return rc;
    if (!e->fused_fast && (rc = e->d_score.ensure(B * g.frame_bytes))) return rc;
    if ((rc = e->d_cand.ensure(std::max<size_t>(B * g.cand_per_frame * 4, 16)))) return rc;
    if ((rc = e->d_ntotal.ensure(B * g.cells_per_frame * 4))) return rc;
    if ((rc = e->d_nretain.ensure(B * g.cells_per_frame * 4))) return rc;
    // HARRIS_SCORE: 64-bit keys (float response above the 32-bit key) for the selection and its output
    const size_t key_bytes = e->scoreType == ORBHIP_HARRIS_SCORE ? 8 : 4;
    if (key_bytes == 8 && (rc = e->d_cand64.ensure(std::max<size_t>(B * g.cand_per_frame * 8, 16)))) return rc;
    if ((rc = e->d_lvl.ensure(std::max<size_t>(B * g.lvl_per_frame * key_bytes, 16)))) return rc;
    if ((rc = e->d_lvlcount.ensure(B * g.nlevels * 4))) return rc;
    if ((rc = e->d_angles.ensure(B * g.nfeatures * 4))) return rc;
    if ((rc = e->d_slots.ensure(B * g.nfeatures * orbk::slot_record_bytes()))) return rc;
    return ORBHIP_OK;
}

// The kernel sequence for `batch` frames whose images, outputs and count array are in device memory.
// `slot0` is the arena slot of the first frame; all per-frame arenas are offset by it.
// `lane` names the stream: 0 = e->stream, 1 + i = e->workers[i].
int run_chunk(orbhip_extractor* e, hipStream_t st, int lane, int slot0, const uint8_t* d_imgs, int batch, long long row_stride,
              long long frame_stride, OrbKeypoint* d_kp, uint8_t* d_desc, int cap, int* d_n) {
    const GeomDev& g = e->hg;
    const GeomDev* dg = e->d_geom.as<GeomDev>();
    uint8_t* pyr = e->d_pyr.as<uint8_t>() + (size_t)slot0 * g.frame_bytes;
    uint8_t* blur = e->d_blur.as<uint8_t>() + (size_t)slot0 * g.frame_bytes;
    uint8_t* score = e->fused_fast ? nullptr : e->d_score.as<uint8_t>() + (size_t)slot0 * g.frame_bytes;
    uint32_t* cand = e->d_cand.as<uint32_t>() + (size_t)slot0 * g.cand_per_frame;
    int* ntotal = e->d_ntotal.as<int>() + (size_t)slot0 * g.cells_per_frame;
    int* nretain = e->d_nretain.as<int>() + (size_t)slot0 * g.cells_per_frame;
    const bool harris = e->scoreType == ORBHIP_HARRIS_SCORE;
    const int key_words = harris ? 2 : 1;
    uint32_t* lvl = e->d_lvl.as<uint32_t>() + (size_t)slot0 * g.lvl_per_frame * key_words;
    int* lvlcount = e->d_lvlcount.as<int>() + (size_t)slot0 * g.nlevels;
    float* angles = e->d_angles.as<float>() + (size_t)slot0 * g.nfeatures;
    hipEvent_t* pev = nullptr;
    if (e->profiling) {
        const size_t need = (size_t)(e->prof_chunks + 1) * (ORB_NSTAGES + 1);
        while (e->prof_ev.size() < need) {
            hipEvent_t ev;
            HIPCHK(hipEventCreate(&ev));
            e->prof_ev.push_back(ev);
        }
        pev = &e->prof_ev[(size_t)e->prof_chunks * (ORB_NSTAGES + 1)];
        e->prof_chunks++;
    }
#define STAGE_MARK(i) do { if (pev) HIPCHK(hipEventRecord(pev[i], st)); } while (0)
    STAGE_MARK(0);
    orbk::launch_level0(st, g, batch, d_imgs, frame_stride, row_stride, pyr, dg);
    STAGE_MARK(1);
    for (int l = 1; l < g.nlevels; l++)
        orbk::launch_resize(st, g, batch, l, pyr, dg, e->d_xofs.as<int>(), e->d_xab.as<uint32_t>(),
                            e->d_ysrc.as<uint32_t>(), e->d_yab.as<uint32_t>());
    STAGE_MARK(2);
    if (!e->fused_fast)
        for (int l = 0; l < g.nlevels; l++) orbk::launch_score(st, g, batch, l, pyr, score, dg);
    STAGE_MARK(3);
    if (e->fused_fast)
        orbk::launch_fast_cell(st, g, batch, e->max_cell_w, e->max_cell_h, pyr, e->d_fcrecs.p, cand, ntotal, e->d_err.as<int>());
    else
        orbk::launch_nms(st, g, batch, e->max_cell_h, score, dg, e->d_cells.as<CellDesc>(), cand, ntotal, e->d_err.as<int>());
    STAGE_MARK(4);
    // The blur needs the pyramid only and is bound by memory latency, the selection kernels by the latency of their
    // LDS round trips: side by side they take less than one after the other (3.59 -> 3.54 ms per 1024 frames of
    // 640 x 480).  Not while stage times are being taken, not inside a captured graph (a fork makes its launch
    // slower), not for small batches.
    hipEvent_t blur_done = nullptr;
    if (e->blur_beside_selection && !e->profiling && !e->in_capture && batch >= 64 && lane < (int)e->blur_streams.size()) {
        hipStream_t bs = e->blur_streams[lane];
        HIPCHK(hipEventRecord(e->blur_fork[lane], st));
        HIPCHK(hipStreamWaitEvent(bs, e->blur_fork[lane], 0));
        orbk::launch_blur(bs, g, batch, pyr, blur, dg, e->gtaps, e->gauss_tie_even);
        blur_done = e->blur_join[lane];
        HIPCHK(hipEventRecord(blur_done, bs));
    }
    if (harris)
        orbk::launch_harris_select(st, g, batch, pyr, dg, e->d_cells.as<CellDesc>(), cand,
                                   e->d_cand64.as<uint64_t>() + (size_t)slot0 * g.cand_per_frame, ntotal, nretain,
                                   (uint64_t*)lvl, lvlcount, e->d_err.as<int>());
    else
        orbk::launch_select(st, g, batch, dg, e->d_cells.as<CellDesc>(), cand, ntotal, nretain, lvl, lvlcount, e->d_err.as<int>());
    STAGE_MARK(5);
    STAGE_MARK(6);  // the orientation is computed by the descriptor kernel
    if (!blur_done) orbk::launch_blur(st, g, batch, pyr, blur, dg, e->gtaps, e->gauss_tie_even);
    STAGE_MARK(7);
    orbk::launch_orient_describe(st, g, batch, pyr, blur, dg, lvl, key_words, lvlcount, e->d_disc.as<uint32_t>(), angles,
                                 e->d_pattern.as<int8_t>(), e->d_slots.as<uint8_t>() + (size_t)slot0 * g.nfeatures * orbk::slot_record_bytes(),
                                 (OrbKeypointDev*)d_kp, d_desc, d_n, cap, blur_done);
    STAGE_MARK(8);
#undef STAGE_MARK
    HIPCHK(hipGetLastError());
    return ORBHIP_OK;
}

int check_call(orbhip_extractor* e, int batch, int rows, int cols, int cap) {
    if (!e) return orb_fail(ORBHIP_EINVAL, "null extractor");
    if (batch < 0 || rows < 0 || cols < 0) return orb_fail(ORBHIP_EINVAL, "negative size");
    if (cap < e->nfeatures) return orb_fail(ORBHIP_ECAP, "cap %d < nfeatures %d", cap, e->nfeatures);
    return ORBHIP_OK;
}

int check_device_error(orbhip_extractor* e) {
    int herr = 0;
    HIPCHK(hipMemcpy(&herr, e->d_err.p, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) {
        hipMemset(e->d_err.p, 0, sizeof(int));
        return orb_fail(ORBHIP_ECAP, "device arena overflow (flags %d)", herr);
    }
    return ORBHIP_OK;
}

}  // namespace

extern "C" {

int orbhip_extractor_create(orbhip_extractor** out, int nfeatures, float scale_factor, int nlevels, int score_type,
                            int fast_th, int device, int max_width, int max_height, int max_batch) {
    if (!out) return orb_fail(ORBHIP_EINVAL, "null out pointer");
    *out = nullptr;
    if (nfeatures < 1 || nlevels < 1 || nlevels > ORBHIP_MAX_LEVELS || !(scale_factor > 1.0f) || max_batch < 1 ||
        max_width < 1 || max_height < 1 || (score_type != ORBHIP_FAST_SCORE && score_type != ORBHIP_HARRIS_SCORE))
        return orb_fail(ORBHIP_EINVAL, "bad extractor parameters");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return orb_fail(ORBHIP_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return orb_fail(ORBHIP_EINVAL, "device %d of %d", device, ndev);
    HIPCHK(hipSetDevice(device));
    orbhip_extractor* e = new orbhip_extractor;
    e->nfeatures = nfeatures;
    e->scaleFactor = scale_factor;  // double member initialised from a float, include/ORBextractor.h:62
    e->nlevels = nlevels;
    e->scoreType = score_type;
    e->fastTh = fast_th;
    e->device = device;
    e->max_batch = max_batch;
    build_tables(e);
    int rc = ORBHIP_OK;
    do {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
            rc = orb_fail(ORBHIP_EHIP, "stream/event creation failed");
            break;
        }
        {
            const char* gr = getenv("ORBHIP_GRAPH");  // 0: never capture the one-frame sequence as a graph
            e->graph_off = gr && gr[0] == '0';
        }
        {
            const char* env = getenv("ORBHIP_STREAMS");
            int nw = env ? atoi(env) : 2;
            nw = std::max(1, std::min(nw, 16));
            bool ok = true;
            for (int i = 0; i < nw && ok; i++) {
                hipStream_t st;
                ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
                if (ok) e->workers.push_back(st);
            }
            for (int i = 0; i < nw + 1 && ok; i++) {
                hipEvent_t ev;
                ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
                if (ok) e->worker_done.push_back(ev);
            }
            const char* bs = getenv("ORBHIP_BLUR_STREAM");  // 0: the blur stays in line with the other kernels
            e->blur_beside_selection = !(bs && bs[0] == '0');
            for (int i = 0; i < nw + 1 && ok && e->blur_beside_selection; i++) {
                hipStream_t st;
                hipEvent_t a = nullptr, b = nullptr;
                ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
                if (ok) e->blur_streams.push_back(st);
                ok = ok && hipEventCreateWithFlags(&a, hipEventDisableTiming) == hipSuccess;
                if (a) e->blur_fork.push_back(a);
                ok = ok && hipEventCreateWithFlags(&b, hipEventDisableTiming) == hipSuccess;
                if (b) e->blur_join.push_back(b);
            }
            if (!ok) {
                rc = orb_fail(ORBHIP_EHIP, "worker stream creation failed");
                break;
            }
        }
        // orientation table: per (disc row v, dword j) the byte weights u + 16 (0 outside the disc) and
        // the inside mask of the four pixels u = -15 + 4j .. -12 + 4j
        std::vector<uint32_t> otab(2 * 256, 0u);
        int inside_total = 0;
        for (int r = 0; r < 31; r++)
            for (int j = 0; j < 8; j++) {
                const int v = r - ORB_HALF_PATCH, d = e->umax[v < 0 ? -v : v];
                uint32_t wu = 0, mk = 0;
                for (int b = 0; b < 4; b++) {
                    const int u = -ORB_HALF_PATCH + 4 * j + b;
                    if (u >= -d && u <= d) {
                        wu |= (uint32_t)(u + 16) << (8 * b);
                        mk |= 1u << (8 * b);
                        inside_total++;
                    }
                }
                otab[2 * (r * 8 + j)] = wu;
                otab[2 * (r * 8 + j) + 1] = mk;
            }
        if (inside_total != ORB_DISC_PIXELS) {
            rc = orb_fail(ORBHIP_EINVAL, "disc has %d pixels", inside_total);
            break;
        }
        if ((rc = upload(e->d_disc, otab.data(), otab.size() * sizeof(uint32_t), e->stream))) break;
        if ((rc = upload(e->d_pattern, kPattern, sizeof(kPattern), e->stream))) break;
        if ((rc = e->d_err.ensure(16))) break;
        if (hipMemset(e->d_err.p, 0, 16) != hipSuccess) {
            rc = orb_fail(ORBHIP_EHIP, "hipMemset failed");
            break;
        }
        if ((rc = set_size(e, max_height, max_width, max_batch))) break;
    } while (0);
    if (rc) {
        orbhip_extractor_destroy(e);
        return rc;
    }
    *out = e;
    return ORBHIP_OK;
}

void orbhip_extractor_destroy(orbhip_extractor* e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    DeviceBuf* bufs[] = {&e->d_geom, &e->d_cells, &e->d_fcrecs, &e->d_xofs, &e->d_xab, &e->d_ysrc, &e->d_yab, &e->d_disc, &e->d_pattern,
                         &e->d_pyr, &e->d_blur, &e->d_score, &e->d_cand, &e->d_cand64, &e->d_ntotal, &e->d_nretain, &e->d_lvl,
                         &e->d_lvlcount, &e->d_angles, &e->d_slots, &e->d_err, &e->d_img, &e->d_outblk};
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
    for (DeviceBuf* b : bufs) b->release();
    e->h_img.release();
    e->h_outblk.release();
    for (hipEvent_t ev : e->prof_ev) hipEventDestroy(ev);
    for (hipEvent_t ev : e->worker_done) hipEventDestroy(ev);
    for (hipStream_t st : e->blur_streams) {
        hipStreamSynchronize(st);
        hipStreamDestroy(st);
    }
    for (hipEvent_t ev : e->blur_fork) hipEventDestroy(ev);
    for (hipEvent_t ev : e->blur_join) hipEventDestroy(ev);
    for (hipStream_t st : e->workers) hipStreamDestroy(st);
    for (hipEvent_t ev : e->copy_done) hipEventDestroy(ev);
    if (e->copy_stream) hipStreamDestroy(e->copy_stream);
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

int orbhip_levels(const orbhip_extractor* e) { return e ? e->nlevels : ORBHIP_EINVAL; }
float orbhip_scale_factor(const orbhip_extractor* e) { return e ? (float)e->scaleFactor : 0.f; }

int orbhip_table(const orbhip_extractor* e, int which, void* out) {
    if (!e || !out) return ORBHIP_EINVAL;
    switch (which) {
        case 0: memcpy(out, e->featuresPerLevel.data(), sizeof(int) * e->nlevels); return e->nlevels;
        case 1: memcpy(out, e->scale.data(), sizeof(float) * e->nlevels); return e->nlevels;
        case 2: memcpy(out, e->invScale.data(), sizeof(float) * e->nlevels); return e->nlevels;
        case 3: memcpy(out, e->umax, sizeof(int) * 16); return 16;
        case 4: memcpy(out, e->gtaps, sizeof(int) * 7); return 7;
    }
    return ORBHIP_EINVAL;
}

int orbhip_extract_batch_device(orbhip_extractor* e, const uint8_t* d_imgs, int batch, int rows, int cols,
                                ptrdiff_t row_stride, ptrdiff_t frame_stride, OrbKeypoint* d_kp_out,
                                uint8_t* d_desc_out, int cap, int* d_n_out) {
    int rc = check_call(e, batch, rows, cols, cap);
    if (rc) return rc;
    if (batch == 0 || rows == 0 || cols == 0) return ORBHIP_OK;  // empty image: silent no-op
    if (!d_imgs || !d_kp_out || !d_desc_out || !d_n_out) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (row_stride < cols) return orb_fail(ORBHIP_EINVAL, "row stride %td < cols %d", row_stride, cols);
    HIPCHK(hipSetDevice(e->device));
    const int chunk = std::min(batch, e->max_batch);
    if ((rc = set_size(e, rows, cols, chunk))) return rc;
    e->prof_chunks = 0;
    (void)hipGetLastError();  // a stale error of an earlier, unrelated call must not be charged to these launches
    if (!e->in_capture) HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (int f0 = 0; f0 < batch; f0 += chunk) {
        const int b = std::min(chunk, batch - f0);
        // sub-batches on the worker streams (single stream while per-stage timing is on, or for tiny chunks)
        const int nsub = e->profiling ? 1 : std::max(1, std::min<int>((int)e->workers.size(), b / 8));
        if (nsub == 1) {
            rc = run_chunk(e, e->stream, 0, 0, d_imgs + (long long)f0 * frame_stride, b, row_stride, frame_stride,
                           d_kp_out + (long long)f0 * cap, d_desc_out + (long long)f0 * cap * 32, cap, d_n_out + f0);
            if (rc) return rc;
        } else {
            hipEvent_t start = e->worker_done[nsub];  // spare event: the chunk may start once earlier work is done
            HIPCHK(hipEventRecord(start, e->stream));
            for (int sub = 0; sub < nsub; sub++) {
                const int s0 = (int)((long long)b * sub / nsub), s1 = (int)((long long)b * (sub + 1) / nsub);
                hipStream_t st = e->workers[sub];
                HIPCHK(hipStreamWaitEvent(st, start, 0));
                rc = run_chunk(e, st, 1 + sub, s0, d_imgs + (long long)(f0 + s0) * frame_stride, s1 - s0, row_stride, frame_stride,
                               d_kp_out + (long long)(f0 + s0) * cap, d_desc_out + (long long)(f0 + s0) * cap * 32, cap,
                               d_n_out + f0 + s0);
                if (rc) return rc;
                HIPCHK(hipEventRecord(e->worker_done[sub], st));
                HIPCHK(hipStreamWaitEvent(e->stream, e->worker_done[sub], 0));
            }
        }
        e->last_batch = b;
    }
    if (!e->in_capture) HIPCHK(hipEventRecord(e->ev1, e->stream));
    return ORBHIP_OK;
}

// Waits for the handle's stream and reads the event timings of the last call.
static int wait_and_collect_times(orbhip_extractor* e) {
    if (!e) return orb_fail(ORBHIP_EINVAL, "null extractor");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (hipEventElapsedTime(&e->last_ms, e->ev0, e->ev1) != hipSuccess) {
        e->last_ms = 0.f;
        (void)hipGetLastError();
    }
    if (e->profiling) {
        const int nl = e->nlevels;
        const int per_chunk[ORB_NSTAGES] = {1, nl - 1, e->fused_fast ? 0 : nl, 1, e->scoreType == ORBHIP_HARRIS_SCORE ? 4 : 3, 0, 1, 2};
        for (int s = 0; s < ORB_NSTAGES; s++) {
            e->stage_ms[s] = 0.f;
            e->stage_launches[s] = per_chunk[s] * e->prof_chunks;
        }
        for (int c = 0; c < e->prof_chunks; c++)
            for (int s = 0; s < ORB_NSTAGES; s++) {
                float ms = 0.f;
                hipEvent_t* pev = &e->prof_ev[(size_t)c * (ORB_NSTAGES + 1)];
                if (hipEventElapsedTime(&ms, pev[s], pev[s + 1]) == hipSuccess) e->stage_ms[s] += ms;
            }
    }
    return ORBHIP_OK;
}

int orbhip_extractor_synchronize(orbhip_extractor* e) {
    const int rc = wait_and_collect_times(e);
    return rc ? rc : check_device_error(e);
}

int orbhip_extractor_set_gaussian_profile(orbhip_extractor* e, int profile) {
    if (!e) return orb_fail(ORBHIP_EINVAL, "null extractor");
    if (profile != ORBHIP_GAUSS_SCALAR && profile != ORBHIP_GAUSS_SSE2) return orb_fail(ORBHIP_EINVAL, "unknown profile %d", profile);
    e->gauss_tie_even = profile == ORBHIP_GAUSS_SSE2;
    return ORBHIP_OK;
}

int orbhip_extractor_set_profiling(orbhip_extractor* e, int enable) {
    if (!e) return orb_fail(ORBHIP_EINVAL, "null extractor");
    e->profiling = enable != 0;
    return ORBHIP_OK;
}

int orbhip_extractor_stage_times(orbhip_extractor* e, float* ms_out, int* launches_out) {
    if (!e || !ms_out || !launches_out) return orb_fail(ORBHIP_EINVAL, "null argument");
    for (int s = 0; s < ORB_NSTAGES; s++) {
        ms_out[s] = e->stage_ms[s];
        launches_out[s] = e->stage_launches[s];
    }
    return ORB_NSTAGES;
}

void* orbhip_extractor_stream(orbhip_extractor* e) { return e ? (void*)e->stream : nullptr; }
float orbhip_last_elapsed_ms(orbhip_extractor* e) { return e ? e->last_ms : 0.f; }

// true if the byte at p lies in page-locked host memory the device can copy from directly (hipHostMalloc,
// hipHostRegister, a pinned tensor); pageable memory reports "unregistered" or an error, depending on the runtime
static bool host_page_locked(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int orbhip_extract_batch(orbhip_extractor* e, const uint8_t* imgs, int batch, int rows, int cols, ptrdiff_t row_stride,
                         ptrdiff_t frame_stride, OrbKeypoint* kp_out, uint8_t* desc_out, int cap, int* n_out) {
    int rc = check_call(e, batch, rows, cols, cap);
    if (rc) return rc;
    if (batch == 0 || rows == 0 || cols == 0) return ORBHIP_OK;
    if (!imgs || !kp_out || !desc_out || !n_out) return orb_fail(ORBHIP_EINVAL, "null buffer");
    if (row_stride < cols) return orb_fail(ORBHIP_EINVAL, "row stride %td < cols %d", row_stride, cols);
    HIPCHK(hipSetDevice(e->device));
    const int chunk = std::min(batch, e->max_batch);
    const size_t img_bytes = (size_t)rows * cols;
    // output block of a chunk: [device error flags, 16 B][counts][keypoints][descriptors], each part 256-byte aligned
    const size_t n_off = 256, kp_off = n_off + (((size_t)chunk * sizeof(int) + 255) & ~(size_t)255);
    const size_t desc_off = kp_off + ((sizeof(OrbKeypoint) * (size_t)cap * chunk + 255) & ~(size_t)255);
    const size_t blk_bytes = desc_off + (size_t)32 * cap * chunk;
    if ((rc = e->d_img.ensure(img_bytes * chunk))) return rc;
    if ((rc = e->d_outblk.ensure(blk_bytes))) return rc;
    if ((rc = e->h_img.ensure(img_bytes * chunk))) return rc;
    if ((rc = e->h_outblk.ensure(blk_bytes))) return rc;
    uint8_t* const dblk = e->d_outblk.as<uint8_t>();
    const uint8_t* const hblk = e->h_outblk.as<uint8_t>();
    // Packed frames in page-locked memory of the caller's are copied from where they lie; anything else goes through
    // the handle's page-locked staging first.  (One frame per call always stages: its queue is a captured graph
    // with the staging address in it.)
    const bool direct = batch > 1 && row_stride == cols && frame_stride == (ptrdiff_t)img_bytes && host_page_locked(imgs) &&
                        host_page_locked(imgs + img_bytes * (size_t)batch - 1);
    for (int f0 = 0; f0 < batch; f0 += chunk) {
        const int b = std::min(chunk, batch - f0);
        // frames -> page-locked staging (packed rows), one host-to-device copy
        const uint8_t* hin = direct ? imgs + (size_t)f0 * img_bytes : e->h_img.as<uint8_t>();
        uint8_t* const stage = e->h_img.as<uint8_t>();
        // Staged frames go to the device in blocks of about 8 MB as soon as a block is packed, so that the copy
        // engine works while the host packs the next block (one frame per call keeps its single copy: it is part
        // of the captured graph).
        const int blk = batch > 1 ? std::max(1, (int)(((size_t)8 << 20) / img_bytes)) : b;
        bool h2d_queued = false;
        const size_t used = desc_off + (size_t)32 * cap * b;
        // Large chunks run as pieces of at least 256 frames: a piece's frames travel on the copy stream, its kernels wait
        // for them alone, so the kernels of one piece run under the copy (and, for staged input, the packing) of the
        // next.  (The stage taps of the debug interface see the last piece only; chunks below 512 frames run whole.)
        const int pieces = e->profiling ? 1 : std::min(8, b / 256);
        bool pipelined = false;
        if (pieces >= 2) {
            if (!e->copy_stream) HIPCHK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
            while ((int)e->copy_done.size() < 8) {
                hipEvent_t ev;
                HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                e->copy_done.push_back(ev);
            }
            // Arenas for the largest piece before the first one is queued: pieces are cut at multiples of eight frames, so
            // a later piece can be larger than the first, and growing an arena then would free memory that queued
            // kernels still use (and serialise the overlap this path exists for).
            auto piece_begin = [&](int k) { return k >= pieces ? b : (int)((long long)b * k / pieces) & ~7; };
            int largest = 0;
            for (int k = 0; k < pieces; k++) largest = std::max(largest, piece_begin(k + 1) - piece_begin(k));
            if ((rc = set_size(e, rows, cols, std::min(largest, e->max_batch)))) return rc;
            // an error inside the loop leaves copies and kernels in flight that read the staging area or the caller's
            // frames: drain both streams before handing the error back
            auto fail = [&](int code) {
                (void)hipStreamSynchronize(e->copy_stream);
                for (hipStream_t st : e->workers) (void)hipStreamSynchronize(st);
                for (hipStream_t st : e->blur_streams) (void)hipStreamSynchronize(st);
                (void)hipStreamSynchronize(e->stream);
                return code;
            };
            for (int k = 0; k < pieces; k++) {
                const int s0 = piece_begin(k), s1 = piece_begin(k + 1);
                const uint8_t* src = imgs + (size_t)(f0 + s0) * img_bytes;
                if (!direct) {
                    // one thread packs about 10 GB/s, a fifth of what the link moves: a piece is packed by a few
                    // threads (ORBHIP_HOST_THREADS, default 4; 1 = the calling thread alone)
                    auto pack = [&](int fa, int fz) {
                        for (int f = fa; f < fz; f++) {
                            const uint8_t* fr = imgs + (ptrdiff_t)(f0 + f) * frame_stride;
                            if (row_stride == cols) {
                                memcpy(stage + (size_t)f * img_bytes, fr, img_bytes);
                            } else {
                                for (int r = 0; r < rows; r++) memcpy(stage + (size_t)f * img_bytes + (size_t)r * cols, fr + (ptrdiff_t)r * row_stride, cols);
                            }
                        }
                    };
                    static const int host_threads = [] {
                        const char* env = getenv("ORBHIP_HOST_THREADS");
                        return std::max(1, std::min(env ? atoi(env) : 4, 16));
                    }();
                    const int nt = std::min(host_threads, s1 - s0);
                    std::vector<std::thread> helpers;
                    for (int t = 1; t < nt; t++) {
                        const int fa = s0 + (int)((long long)(s1 - s0) * t / nt), fz = s0 + (int)((long long)(s1 - s0) * (t + 1) / nt);
                        try {
                            helpers.emplace_back(pack, fa, fz);
                        } catch (...) {
                            pack(fa, fz);  // no thread to be had: the caller's thread does the share itself
                        }
                    }
                    pack(s0, s0 + (int)((long long)(s1 - s0) / nt));
                    for (std::thread& th : helpers) th.join();
                    src = stage + (size_t)s0 * img_bytes;
                }
                if (hipMemcpyAsync(e->d_img.as<uint8_t>() + (size_t)s0 * img_bytes, src, img_bytes * (size_t)(s1 - s0),
                                   hipMemcpyHostToDevice, e->copy_stream) != hipSuccess ||
                    hipEventRecord(e->copy_done[k], e->copy_stream) != hipSuccess ||
                    hipStreamWaitEvent(e->stream, e->copy_done[k], 0) != hipSuccess)
                    return fail(orb_fail(ORBHIP_EHIP, "queueing the copy of piece %d failed: %s", k, hipGetErrorString(hipGetLastError())));
                if ((rc = orbhip_extract_batch_device(e, e->d_img.as<uint8_t>() + (size_t)s0 * img_bytes, s1 - s0, rows, cols, cols,
                                                      (ptrdiff_t)img_bytes, (OrbKeypoint*)(dblk + kp_off) + (size_t)s0 * cap,
                                                      dblk + desc_off + (size_t)32 * cap * s0, cap, (int*)(dblk + n_off) + s0)))
                    return fail(rc);
            }
            HIPCHK(hipMemcpyAsync(dblk, e->d_err.p, sizeof(int), hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemcpyAsync((void*)hblk, dblk, used, hipMemcpyDeviceToHost, e->stream));
            pipelined = true;
        }
        for (int fb = 0; fb < b && !direct && !pipelined; fb += blk) {
            const int nb = std::min(blk, b - fb);
            for (int f = fb; f < fb + nb; f++) {
                const uint8_t* src = imgs + (ptrdiff_t)(f0 + f) * frame_stride;
                if (row_stride == cols) {
                    memcpy(stage + (size_t)f * img_bytes, src, img_bytes);
                } else {
                    for (int r = 0; r < rows; r++) memcpy(stage + (size_t)f * img_bytes + (size_t)r * cols, src + (ptrdiff_t)r * row_stride, cols);
                }
            }
            if (batch > 1) {
                HIPCHK(hipMemcpyAsync(e->d_img.as<uint8_t>() + (size_t)fb * img_bytes, stage + (size_t)fb * img_bytes,
                                      img_bytes * nb, hipMemcpyHostToDevice, e->stream));
                h2d_queued = true;
            }
        }
        // the queue of one chunk: image in, kernel sequence, error flags and the whole output block back (ONE
        // synchronisation; the counts say how much of the block is valid)
        auto enqueue = [&]() -> int {
            if (!h2d_queued) HIPCHK(hipMemcpyAsync(e->d_img.p, hin, img_bytes * b, hipMemcpyHostToDevice, e->stream));
            const int r2 = orbhip_extract_batch_device(e, e->d_img.as<uint8_t>(), b, rows, cols, cols, (ptrdiff_t)img_bytes,
                                                       (OrbKeypoint*)(dblk + kp_off), dblk + desc_off, cap, (int*)(dblk + n_off));
            if (r2) return r2;
            HIPCHK(hipMemcpyAsync(dblk, e->d_err.p, sizeof(int), hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemcpyAsync((void*)hblk, dblk, used, hipMemcpyDeviceToHost, e->stream));
            return ORBHIP_OK;
        };
        bool replayed = false;
        if (batch == 1 && !e->profiling && !e->graph_off) {
            // buffers first (nothing may be allocated while the stream is being captured)
            if ((rc = set_size(e, rows, cols, 1))) return rc;
            orbhip_extractor::GraphKey key;
            key.rows = rows; key.cols = cols; key.cap = cap; key.tie_even = e->gauss_tie_even; key.generation = g_buffer_generation.load();
            if (!e->graph_exec || !(key == e->graph_key)) {
                if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
                e->graph_exec = nullptr;
                HIPCHK(hipStreamSynchronize(e->stream));  // geometry uploads of set_size are not part of the graph
                hipGraph_t graph = nullptr;
                bool ok = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    e->in_capture = true;
                    const int qrc = enqueue();
                    e->in_capture = false;
                    const hipError_t endrc = hipStreamEndCapture(e->stream, &graph);
                    ok = qrc == ORBHIP_OK && endrc == hipSuccess && graph != nullptr && key.generation == g_buffer_generation.load();
                }
                if (ok) ok = hipGraphInstantiate(&e->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph) (void)hipGraphDestroy(graph);
                if (!ok) {
                    (void)hipGetLastError();
                    e->graph_exec = nullptr;
                    e->graph_off = true;
                } else {
                    e->graph_key = key;
                }
            }
            if (e->graph_exec) {
                // the timing events bracket the replay (events recorded inside a capture belong to the graph and
                // cannot be queried)
                HIPCHK(hipEventRecord(e->ev0, e->stream));
                HIPCHK(hipGraphLaunch(e->graph_exec, e->stream));
                HIPCHK(hipEventRecord(e->ev1, e->stream));
                e->last_batch = 1;
                replayed = true;
            }
        }
        if (!replayed && !pipelined && (rc = enqueue())) return rc;
        if ((rc = wait_and_collect_times(e))) return rc;
        const int herr = *(const int*)hblk;
        if (herr) {
            hipMemset(e->d_err.p, 0, sizeof(int));
            return orb_fail(ORBHIP_ECAP, "device arena overflow (flags %d)", herr);
        }
        const int* hn = (const int*)(hblk + n_off);
        for (int f = 0; f < b; f++) {
            const int n = hn[f];
            n_out[f0 + f] = n;
            if (n > 0) {
                memcpy(kp_out + (size_t)(f0 + f) * cap, hblk + kp_off + sizeof(OrbKeypoint) * (size_t)f * cap, sizeof(OrbKeypoint) * n);
                memcpy(desc_out + (size_t)(f0 + f) * cap * 32, hblk + desc_off + (size_t)32 * f * cap, (size_t)32 * n);
            }
        }
    }
    return ORBHIP_OK;
}

int orbhip_extract(orbhip_extractor* e, const uint8_t* img, int rows, int cols, ptrdiff_t row_stride, OrbKeypoint* kp_out,
                   uint8_t* desc_out, int cap, int* n_out) {
    return orbhip_extract_batch(e, img, 1, rows, cols, row_stride, 0, kp_out, desc_out, cap, n_out);
}

int orbhip_level_geometry(const orbhip_extractor* e, int level, int32_t* out8) {
    if (!e || !out8 || level < 0 || level >= e->nlevels) return ORBHIP_EINVAL;
    const LevelGeom& L = e->hg.lv[level];
    out8[0] = L.w; out8[1] = L.h; out8[2] = L.cols; out8[3] = L.rows;
    out8[4] = L.cellW; out8[5] = L.cellH; out8[6] = L.nfeaturesCell; out8[7] = L.quota;
    return ORBHIP_OK;
}

int orbhip_debug_get(orbhip_extractor* e, int stage, int frame, int level, void* buf, size_t buf_bytes) {
    if (!e || !buf || level < 0 || level >= e->nlevels || frame < 0 || frame >= e->last_batch)
        return orb_fail(ORBHIP_EINVAL, "bad tap request");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const GeomDev& g = e->hg;
    const LevelGeom& L = g.lv[level];
    if (stage == 2 && e->fused_fast) return orb_fail(ORBHIP_EUNSUPPORTED, "no measure plane: the fused FAST kernel keeps it in LDS");
    if (stage >= 0 && stage <= 2) {
        const int pw = L.w + 2 * ORB_EDGE;
        if (buf_bytes < (size_t)pw * L.ph) return orb_fail(ORBHIP_ECAP, "tap buffer too small");
        const DeviceBuf& src = stage == 0 ? e->d_pyr : stage == 1 ? e->d_blur : e->d_score;
        HIPCHK(hipMemcpy2D(buf, pw, src.as<uint8_t>() + (size_t)frame * g.frame_bytes + L.plane_off, L.stride, pw, L.ph,
                           hipMemcpyDeviceToHost));
        return pw * L.ph;
    }
    if (stage == 3 || stage == 4) {
        if (buf_bytes < sizeof(int) * (size_t)L.nCells) return orb_fail(ORBHIP_ECAP, "tap buffer too small");
        const DeviceBuf& src = stage == 3 ? e->d_ntotal : e->d_nretain;
        HIPCHK(hipMemcpy(buf, src.as<int>() + (size_t)frame * g.cells_per_frame + L.cell_base, sizeof(int) * L.nCells,
                         hipMemcpyDeviceToHost));
        return L.nCells;
    }
    if (stage == 5) {
        int count = 0;
        HIPCHK(hipMemcpy(&count, e->d_lvlcount.as<int>() + (size_t)frame * g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
        if (buf_bytes < sizeof(int) * 3 * (size_t)count) return orb_fail(ORBHIP_ECAP, "tap buffer too small");
        const int kw = e->scoreType == ORBHIP_HARRIS_SCORE ? 2 : 1;
        std::vector<uint32_t> keys((size_t)count * kw);
        if (count)
            HIPCHK(hipMemcpy(keys.data(), e->d_lvl.as<uint32_t>() + ((size_t)frame * g.lvl_per_frame + L.lvl_off) * kw,
                             sizeof(uint32_t) * count * kw, hipMemcpyDeviceToHost));
        int* o = (int*)buf;
        for (int i = 0; i < count; i++) {
            const uint32_t key = keys[(size_t)i * kw];
            o[3 * i] = key & 0xfff;
            o[3 * i + 1] = (key >> 12) & 0xfff;
            o[3 * i + 2] = kw == 2 ? (int)ordered_bits_to_float_bits(keys[(size_t)i * kw + 1]) : (int)(key >> 24);
        }
        return count;
    }
    if (stage == 6) {
        std::vector<int> counts(L.nCells);
        HIPCHK(hipMemcpy(counts.data(), e->d_ntotal.as<int>() + (size_t)frame * g.cells_per_frame + L.cell_base,
                         sizeof(int) * L.nCells, hipMemcpyDeviceToHost));
        size_t total = 0;
        for (int c = 0; c < L.nCells; c++) total += counts[c];
        if (buf_bytes < sizeof(int) * 3 * total) return orb_fail(ORBHIP_ECAP, "tap buffer too small");
        int* o = (int*)buf;
        const int kw = e->scoreType == ORBHIP_HARRIS_SCORE ? 2 : 1;
        const DeviceBuf& arena = kw == 2 ? e->d_cand64 : e->d_cand;
        std::vector<uint32_t> keys;
        for (int c = 0; c < L.nCells; c++) {
            const CellDesc& cd = e->hcells[L.cell_base + c];
            keys.resize((size_t)counts[c] * kw);
            if (counts[c])
                HIPCHK(hipMemcpy(keys.data(), arena.as<uint32_t>() + ((size_t)frame * g.cand_per_frame + cd.cand_off) * kw,
                                 sizeof(uint32_t) * counts[c] * kw, hipMemcpyDeviceToHost));
            for (int i = 0; i < counts[c]; i++) {
                const uint32_t key = keys[(size_t)i * kw];
                *o++ = key & 0xfff;
                *o++ = (key >> 12) & 0xfff;
                *o++ = kw == 2 ? (int)ordered_bits_to_float_bits(keys[(size_t)i * kw + 1]) : (int)(key >> 24);
            }
        }
        return (int)total;
    }
    return orb_fail(ORBHIP_EINVAL, "unknown tap stage %d", stage);
}

// Device self-tests of arithmetic helpers (test instrumentation; declared in orbhip_test.h).
int orbhip_selftest(int which, const void* in, int n, int aux, void* out) {
    if (n <= 0) return ORBHIP_OK;
    if ((which == 4 || which == 5) && n > 4096) return orb_fail(ORBHIP_EINVAL, "wave select selftest holds at most 4096 keys");
    size_t in_bytes, out_bytes;
    switch (which) {
        case 0: in_bytes = 8ull * n; out_bytes = 4ull * n; break;   // atan2: (y, x) pairs -> degrees
        case 1: in_bytes = 4ull * n; out_bytes = 8ull * n; break;   // angle -> (cos, sin)
        case 2: in_bytes = 4ull * n; out_bytes = 4ull * n; break;   // keys, nth = aux -> permuted keys
        case 3: in_bytes = 49ull * n; out_bytes = 4ull * n; break;  // 7x7 patches -> corner measure
        case 4: in_bytes = 4ull * n; out_bytes = 4ull * n; break;   // keys, wave-cooperative nth_element (n <= 4096)
        case 5: in_bytes = 4ull * n; out_bytes = 4ull * n; break;   // the same with depth limit 0 (heap-select branch)
        default: return orb_fail(ORBHIP_EINVAL, "unknown selftest");
    }
    void *din = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc(&din, in_bytes));
    if (hipMalloc(&dout, out_bytes) != hipSuccess) {
        hipFree(din);
        return orb_fail(ORBHIP_ENOMEM, "hipMalloc failed");
    }
    hipMemcpy(din, in, in_bytes, hipMemcpyHostToDevice);
    switch (which) {
        case 0: orbk::launch_test_atan2(0, (const float*)din, (float*)dout, n); break;
        case 1: orbk::launch_test_sincos(0, (const float*)din, (float*)dout, n); break;
        case 2:
            orbk::launch_test_select(0, (uint32_t*)din, n, aux);
            break;
        case 3: orbk::launch_test_measure(0, (const uint8_t*)din, (int*)dout, n); break;
        case 4: orbk::launch_test_wave_select(0, (uint32_t*)din, n, aux, -1); break;
        case 5: orbk::launch_test_wave_select(0, (uint32_t*)din, n, aux, 1); break;
    }
    hipError_t err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemcpy(out, (which == 2 || which == 4 || which == 5) ? din : dout, out_bytes, hipMemcpyDeviceToHost);
    hipFree(din);
    hipFree(dout);
    if (err != hipSuccess) return orb_fail(ORBHIP_EHIP, "selftest failed: %s", hipGetErrorString(err));
    return ORBHIP_OK;
}

}  // extern "C"
