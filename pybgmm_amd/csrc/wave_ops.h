// Wave-wide (64 lanes) reductions and scans that stay off the LDS crossbar: all-reduce inside each
// row of 16 lanes with DPP, the four row results through v_readlane.  For kernels whose cost is the
// latency of one dependent chain (the sequential parts of the sampler).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ double wv_readlane(double v, int t) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), t);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), t);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ long long wv_readlane_i64(long long v, int t) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), t);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), t);
    return ((long long)hi << 32) | (unsigned int)lo;
}

template <int CTRL>
__device__ __forceinline__ double wv_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wv_max(double v) {
    v = fmax(v, wv_dpp<0xB1>(v));      // quad_perm [1,0,3,2]
    v = fmax(v, wv_dpp<0x4E>(v));      // quad_perm [2,3,0,1]
    v = fmax(v, wv_dpp<0x141>(v));     // row_half_mirror
    v = fmax(v, wv_dpp<0x140>(v));     // row_mirror
    return fmax(fmax(wv_readlane(v, 0), wv_readlane(v, 16)), fmax(wv_readlane(v, 32), wv_readlane(v, 48)));
}

__device__ __forceinline__ double wv_sum(double v) {
    v += wv_dpp<0xB1>(v);
    v += wv_dpp<0x4E>(v);
    v += wv_dpp<0x141>(v);
    v += wv_dpp<0x140>(v);
    return (wv_readlane(v, 0) + wv_readlane(v, 16)) + (wv_readlane(v, 32) + wv_readlane(v, 48));
}

// inclusive prefix sum over the 64 lanes (row_shr 1, 2, 4, 8 inside a row; row totals via v_readlane)
__device__ __forceinline__ double wv_scan(double v, int lane) {
    v += wv_dpp<0x111>(v);
    v += wv_dpp<0x112>(v);
    v += wv_dpp<0x114>(v);
    v += wv_dpp<0x118>(v);
    const double t0 = wv_readlane(v, 15), t1 = wv_readlane(v, 31), t2 = wv_readlane(v, 47);
    const double t01 = t0 + t1;
    return v + (lane < 16 ? 0.0 : (lane < 32 ? t0 : (lane < 48 ? t01 : t01 + t2)));
}
