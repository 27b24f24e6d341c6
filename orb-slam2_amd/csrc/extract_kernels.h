// Launch interface of the extractor kernels (extract_kernels.hip).
#ifndef ORBHIP_EXTRACT_KERNELS_H
#define ORBHIP_EXTRACT_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "geom.h"

struct OrbKeypointDev {  // same 28-byte layout as OrbKeypoint of the C ABI
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

namespace orbk {
void launch_level0(hipStream_t st, const GeomDev& hg, int batch, const uint8_t* imgs, long long frame_stride,
                   long long row_stride, uint8_t* pyr, const GeomDev* dg);
void launch_resize(hipStream_t st, const GeomDev& hg, int batch, int level, uint8_t* pyr, const GeomDev* dg,
                   const int* xofs, const uint32_t* xab, const uint32_t* ysrc, const uint32_t* yab);
void launch_score(hipStream_t st, const GeomDev& hg, int batch, int level, const uint8_t* pyr, uint8_t* score,
                  const GeomDev* dg);
void launch_nms(hipStream_t st, const GeomDev& hg, int batch, int max_cell_h, const uint8_t* score, const GeomDev* dg,
                const CellDesc* cells, uint32_t* cand, int* n_total, int* err);
// Fused per-cell FAST (fast_cell.hip); usable when every cell satisfies fast_cell_supported.
bool fast_cell_supported(int cw, int ch);
// build_fast_cell_recs fills one record of fast_cell_rec_bytes() bytes per cell (host side, once per image size)
size_t fast_cell_rec_bytes();
void build_fast_cell_recs(const GeomDev& hg, const CellDesc* cells, int ncells, void* out);
void launch_fast_cell(hipStream_t st, const GeomDev& hg, int batch, int max_cw, int max_ch, const uint8_t* pyr,
                      const void* recs, uint32_t* cand, int* n_total, int* err);
void launch_select(hipStream_t st, const GeomDev& hg, int batch, const GeomDev* dg, const CellDesc* cells,
                   uint32_t* cand, const int* n_total, int* n_retain, uint32_t* lvl_list, int* lvl_count, int* err);
void launch_blur(hipStream_t st, const GeomDev& hg, int batch, const uint8_t* pyr, uint8_t* blur, const GeomDev* dg,
                 const int taps[7], bool tie_even);
// IC_Angle + computeOrbDescriptor + output records, one half-wave per keypoint.
// lvl_list holds key_words (1 or 2) 32-bit words per keypoint, low word first.
void launch_orient_describe(hipStream_t st, const GeomDev& hg, int batch, const uint8_t* pyr, const uint8_t* blur,
                            const GeomDev* dg, const uint32_t* lvl_list, int key_words, const int* lvl_count,
                            const uint32_t* otab, float* angles, const int8_t* pattern, void* slot_recs,
                            OrbKeypointDev* kp_out, uint8_t* desc_out, int* n_out, int cap, hipEvent_t blur_done = nullptr);
// slot_recs: scratch of nfeatures * slot_record_bytes() per frame
size_t slot_record_bytes();
// HARRIS_SCORE: Harris responses of the FAST candidates into 64-bit keys, then the selection on those.
void launch_harris_select(hipStream_t st, const GeomDev& hg, int batch, const uint8_t* pyr, const GeomDev* dg,
                          const CellDesc* cells, const uint32_t* cand, uint64_t* cand64, const int* n_total, int* n_retain,
                          uint64_t* lvl_list64, int* lvl_count, int* err);
void launch_test_atan2(hipStream_t st, const float* yx, float* out, int n);
void launch_test_sincos(hipStream_t st, const float* in, float* out, int n);
void launch_test_select(hipStream_t st, uint32_t* keys, int n, int nth);
void launch_test_wave_select(hipStream_t st, uint32_t* keys, int n, int nth, int depth);
void launch_test_measure(hipStream_t st, const uint8_t* patches, int* out, int n);
}  // namespace orbk
#endif
