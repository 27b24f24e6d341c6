// Geometry of one extractor configuration at one image size: pyramid level sizes, the cell grid
// of every level and the layout of the per-frame device arenas.  Computed on the host with the
// reference's own float expressions (src/ORBextractor.cc:462-487,527-547,560-599,786) and read by
// every kernel through a pointer to one GeomDev in device memory (wave-uniform scalar loads).
#ifndef ORBHIP_GEOM_H
#define ORBHIP_GEOM_H
#include <stdint.h>

#define ORB_MAX_LEVELS 16
#define ORB_EDGE 16          // EDGE_THRESHOLD, src/ORBextractor.cc:77
#define ORB_HALF_PATCH 15    // HALF_PATCH_SIZE, :76
#define ORB_DISC_PIXELS 749  // pixels of the radius-15 disc the orientation moment runs over
#define ORB_MAX_DIM 4095     // keypoint keys hold 12-bit coordinates
#define ORB_MAX_CELLS_PER_LEVEL 2048

struct LevelGeom {
    int w, h;              // level image (ROI) size
    int stride;            // row stride in bytes of the padded plane, multiple of 64, >= w + 32
    int ph;                // padded height h + 32
    long long plane_off;   // byte offset of the padded plane inside a frame's pyramid
    int cols, rows;        // levelCols, levelRows
    int cellW, cellH;
    int nCells;
    int nfeaturesCell;
    int quota;             // mnFeaturesPerLevel[level]
    int cell_base;         // index of the level's first cell in the per-frame cell table
    int slot_base;         // sum of the quotas of the lower levels = first output slot of the level
    int lvl_cap;           // capacity (keys) of the level's selection list
    long long lvl_off;     // offset (keys) of that list inside a frame's list arena
    int rz_x_off, rz_y_off;  // offsets of the level's resize tables (entries); level 0 unused
    float scale;           // mvScaleFactor[level]
    float kp_size;         // (float)(int)(31 * mvScaleFactor[level])
};

struct GeomDev {
    int nlevels;
    int nfeatures;          // sum of quotas = output slots per frame
    int fast_th;
    int cells_per_frame;
    long long frame_bytes;  // bytes of one frame's pyramid (all padded planes)
    long long cand_per_frame;  // keys of one frame's candidate arena
    long long lvl_per_frame;   // keys of one frame's level-list arena
    LevelGeom lv[ORB_MAX_LEVELS];
};

// One FAST cell: the region of the level image whose pixels the cell's FAST call examines
// ([x0, x0+cw) x [y0, y0+ch) in level coordinates; the cell image starts 3 px before it).
struct CellDesc {
    int level;
    int index;            // i * levelCols + j
    int x0, y0, cw, ch;   // examined pixels; cw <= 0 or ch <= 0: no pixel can be a corner
    int ini_x, ini_y;     // iniXCol[j], iniYRow[i]: cell-image origin = x0 - 3, y0 - 3
    long long cand_off;   // offset (keys) of the cell's candidate list in a frame's arena
    int cand_cap;
    int skipped;          // the reference never visits the cell (:570, :594): hX <= 0 or hY <= 0
};

#ifdef __HIPCC__
// Workgroups are handed to the eight XCDs (each with its own L2) round-robin by linear workgroup
// id.  These helpers permute (inner index, frame) inside each group of eight frames so that XCD k
// runs every workgroup of frame k of the group: workgroups that touch the same image rows then
// share one L2.  `inner` counts the workgroups of one frame; frames past the last full group of
// eight keep the identity mapping.  The permutation is a bijection of the launched grid.
__device__ __forceinline__ void xcd_frame_swizzle(int& inner, int& frame, int inner_count, int frames) {
    if (frame < (frames & ~7)) {
        const uint32_t lin = (uint32_t)(frame & 7) * (uint32_t)inner_count + (uint32_t)inner;
        frame = (frame & ~7) + (int)(lin & 7u);
        inner = (int)(lin >> 3);
    }
}
// 3-D grids (x, y, frame): returns the permuted (bx, by, f).
__device__ __forceinline__ void xcd_frame_swizzle3(int& bx, int& by, int& frame) {
    int inner = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    frame = (int)blockIdx.z;
    xcd_frame_swizzle(inner, frame, (int)(gridDim.x * gridDim.y), (int)gridDim.z);
    by = inner / (int)gridDim.x;
    bx = inner - by * (int)gridDim.x;
}
#endif

#endif
