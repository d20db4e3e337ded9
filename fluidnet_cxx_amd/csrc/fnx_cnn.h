// MultiScale pressure-net (lib/multi_scale_net.py:21-127) -- internal interface of the conv stack.
#pragma once
#include <hip/hip_runtime.h>
#include "fnx_device.h"

namespace fnx {

struct ConvLayer { int cin, cout, k, relu; };

// channel plan, multi_scale_net.py:111-116 (Dropout = identity in eval)
constexpr int N_LAYERS = 17;
constexpr ConvLayer LAYERS[N_LAYERS] = {
    // convN_4 (quarter resolution)
    {2, 32, 3, 1}, {32, 64, 3, 1}, {64, 32, 3, 0}, {32, 1, 3, 0},
    // convN_2 (half resolution)
    {3, 32, 5, 1}, {32, 64, 3, 1}, {64, 128, 3, 1}, {128, 64, 3, 1}, {64, 32, 3, 0}, {32, 1, 3, 0},
    // convN_1 (full resolution)
    {3, 32, 5, 1}, {32, 64, 3, 1}, {64, 128, 3, 1}, {128, 64, 3, 1}, {64, 32, 3, 0}, {32, 8, 5, 0},
    // final 1x1
    {8, 1, 1, 0}};

inline int layer_taps(const ConvLayer& L, bool is3d) { return L.k * L.k * (is3d ? L.k : 1); }
inline size_t layer_weight_floats(const ConvLayer& L, bool is3d) { return (size_t)L.cout * L.cin * layer_taps(L, is3d); }

size_t scalenet_weight_floats(bool is3d);
size_t scalenet_packed_bytes(bool is3d);
void scalenet_pack(bool is3d, const float* blob, void* packed, hipStream_t s);

size_t multiscale_ws_bytes(const GridDims& g, bool is3d);
size_t fluidnet_ws_bytes(const GridDims& g, bool is3d);

// x (B,2,D,H,W) -> p (B,1,D,H,W)
void multiscale_forward(const GridDims& g, bool is3d, const void* packed, const float* x, float* p, int precision_mode, void* ws,
                        hipStream_t s);
// the same on nested z-crops (fnx_cnn.hip): trim = {full-tower low, high, half-tower low, high} in full-resolution planes,
// multiples of 4; p receives g.D - trim[0] - trim[1] planes.  Workspace: multiscale_ws_bytes(g)
void multiscale_forward_crop(const GridDims& g, bool is3d, const void* packed, const float* x, float* p, int precision_mode,
                             void* ws, hipStream_t s, const int trim[4]);

// pieces of FluidNet.forward (lib/model.py:76-227)
size_t scale_std_scratch_bytes(int B);      // `partial` of launch_scale_std (fixed-order fp64 partial sums, no atomics)
void launch_scale_std(const GridDims& g, int nc, const float* U, float thr, double* partial,
                      float* scale /*B*/, hipStream_t s);
void launch_pack_input(const GridDims& g, int nc, const float* div, const float* flags, const float* scale, float* U,
                       float* x, hipStream_t s);
void launch_unscale(const GridDims& g, int nc, const float* scale, float* p, float* U, hipStream_t s);
void launch_gather_input(const GridDims& g, int nc, const float* input, float* U, float* flags, hipStream_t s);

// pieces of the CNN projection on a z-slab (fnx_slab_step with method 1; fnx_slab.hip)
size_t window_sums_scratch_bytes(int B);                       // `partial` of launch_window_sums_encode
// this rank's fp64 (sum, sumsq) of U over the planes [k0, k1) of every channel, as 3 floats per double in ITS slots of
// red[nranks][B][2][3] (all other slots zero): a float all-reduce(sum) over the ranks then is an exact all-gather
void launch_window_sums_encode(const GridDims& g, int nc, int k0, int k1, const float* U, int rank, int nranks, double* partial,
                               float* red, hipStream_t s);
// scale[b] = clamp(unbiased std over n elements, thr) from the gathered sums, added in rank order
void launch_scale_from_sums(int nranks, int B, double n, float thr, const float* red, float* scale, hipStream_t s);
// x[b,0] = div(U / scale[b]), x[b,1] = occupancy(flags) on the planes of g's compute window (U is not modified)
void launch_pack_div(const GridDims& g, bool is3d, const float* U, const float* flags, const float* scale, float* x, hipStream_t s);

}  // namespace fnx

struct FnxGrid;
struct FnxState;
namespace fnx {
// bcs != nullptr: the fused step's form -- the last setConstVals of the step (simulate.py:168) is part of the pass behind the
// net (bcs supplies density, the BC arrays, bc_class and density_bc_applied; p_out must not be the workspace)
int fluidnet_core(const FnxGrid* g, const void* packed, const float* flags, float thr, int precision_mode, float* p_out, float* U,
                  void* ws, void* stream, const FnxState* bcs = nullptr);
}
