// xh_internal.h -- launchers shared between translation units but not part of the public C ABI.
#pragma once
#include "xh_common.h"
#include "../../include/x265hip_frame.h"

int xh_count_nonzero(hipStream_t st, int N, const int16_t* q, int n, uint32_t* out);
int xh_copy_count(hipStream_t st, int N, const int16_t* resi, intptr_t rs, int16_t* coef, uint32_t* out);
int xh_denoise(hipStream_t st, int16_t* coef, uint32_t* resSum, const uint16_t* offset, int num);

// 32x32 forward DCT: MFMA (i8 x i8 -> i32, two byte planes) and VALU/LDS variants
bool xh_dct32_mfma_enabled();
int xh_dct32_mfma(hipStream_t st, const int16_t* src, intptr_t ss, const int32_t* sOff, int16_t* dst, const int32_t* dOff, int n);
int xh_dct16_mfma(hipStream_t st, const int16_t* src, intptr_t ss, const int32_t* sOff, int16_t* dst, const int32_t* dOff, int n);
int xh_idct16_mfma(hipStream_t st, const int16_t* src, const int32_t* sOff, int16_t* dst, intptr_t ds, const int32_t* dOff, int n);
int xh_idct32_mfma(hipStream_t st, const int16_t* src, const int32_t* sOff, int16_t* dst, intptr_t ds, const int32_t* dOff, int n);
int xh_dct32_valu(hipStream_t st, const int16_t* src, intptr_t ss, const int32_t* sOff, int16_t* dst, const int32_t* dOff, int n);

#ifdef X265HIP_EXPERIMENTS
// kern_me_pyr.hip: the 32x32 / 16x16 / 8x8 levels of the CU pyramid of a range of CTU rows in one launch (a wavefront per 32x32 quadrant); the task lists are the batch's
// own (x265hip_batch_build_me_tasks: picture-major raster, mvpFrom = the parent CU)
bool xh_me_pyr_ok(int method, int64_t planeElems, int costHalfRange);
int xh_me_pyr(void* stream, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
              const x265hip_me_task* const* tasks, x265hip_me_result* const* results, const x265hip_me_result* parent64,
              int firstCtuRow, int ctuRows, int width, const uint16_t* costRow, int costHalfRange, int merange, int method, int subpelRefine,
              const void* subpelPlanes, int64_t planeElems);
#endif

// kern_me_star.hip: x265hip_me_batch for 64x64 STAR tasks with zero predictors and no candidates (the batch host's top level) without the start-stage launch
int xh_me_star_own64(void* stream, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                     const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                     int merange, int subpelRefine, x265hip_me_result* results, const void* subpelPlanes, int64_t planeElems);

#ifdef X265HIP_EXPERIMENTS
// Tiled phase planes (16-bit library): producer (kern_planes.hip) and the readers that take them (kern_me_star_tiled.hip, kern_tq.hip)
bool xh_subpel_planes_tiled_ok(intptr_t stride, int rows);
int xh_subpel_planes_tiled(void* stream, const void* refPlane, intptr_t stride, int rows, void* outPlanes, int64_t planeElems);
int xh_me_star_tiled(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                     const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                     int merange, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                     const void* tiledPlanes, int64_t planeElems, bool ownStart64);
int xh_tq_batch_tiled(void* stream, int log2TrSize, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                      const struct x265hip_tu_task* tasks, int n, const struct x265hip_tq_params* params,
                      int16_t* coeff, uint32_t* numSig, void* reconPlane, intptr_t reconStride, uint64_t* sse, const x265hip_me_result* mvSource);
#endif
