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


// kern_me_star.hip: x265hip_me_batch for 64x64 STAR tasks with zero predictors and no candidates (the batch host's top level) without the start-stage launch
int xh_me_star_own64(void* stream, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                     const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                     int merange, int subpelRefine, x265hip_me_result* results, const void* subpelPlanes, int64_t planeElems);

