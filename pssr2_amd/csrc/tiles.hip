// Device-side tiling and reassembly for whole-sheet prediction (BASELINE config 5):
//   * sliding-window LR tiles straight from a uint8 sheet resident in HBM (pssr/data.py:629-638 `_sliding_window`, row-major
//     tiles, stride = size - overlap, trailing remainder dropped; `_tensor_ready`: float32 of the integer values);
//   * overlap-averaged stitching of the predicted uint8 tiles (pssr/util.py:116-137 `_patch_images` followed by the uint8
//     cast of `reassemble_sheets`, pssr/util.py:101): out = floor(sum / count) with `margin` pixels trimmed on inner edges.
// Both are pure index arithmetic + one integer divide per pixel, bit-exact.
#include "common.h"

namespace {

__global__ void sliding_tiles_kernel(const uint8_t* __restrict__ sheet, float* __restrict__ out, int c, int h, int w, int size, int stride,
                                     int tiles_x, int tile0, int ntile) {
    const long total = (long)ntile * c * size * size;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = i % size;
        long t = i / size;
        const int y = t % size; t /= size;
        const int ch = t % c;
        const int tile = tile0 + (int)(t / c);
        const int ty = tile / tiles_x, tx = tile % tiles_x;
        out[i] = (float)sheet[((long)ch * h + ty * stride + y) * w + tx * stride + x];
    }
}

// gather formulation: every output pixel sums the tiles that cover it (after margin trimming) and divides
__global__ void patch_tiles_kernel(const uint8_t* __restrict__ tiles, uint8_t* __restrict__ out, int c, int n_rows, int n_cols, int size,
                                   int overlap, int margin) {
    const int step = size - overlap;
    const int H = n_rows * step + overlap, W = n_cols * step + overlap;
    const long total = (long)c * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = i % W;
        const int y = (i / W) % H;
        const int ch = i / ((long)W * H);
        // tile rows r with r*step <= y < r*step + size
        int r_lo = (y - size) / step + 1; if (r_lo < 0 || y < size) r_lo = (y >= size) ? r_lo : 0;
        int r_hi = y / step; if (r_hi > n_rows - 1) r_hi = n_rows - 1;
        int c_lo = (x - size) / step + 1; if (c_lo < 0 || x < size) c_lo = (x >= size) ? c_lo : 0;
        int c_hi = x / step; if (c_hi > n_cols - 1) c_hi = n_cols - 1;
        unsigned sum = 0, cnt = 0;
        for (int r = r_lo; r <= r_hi; ++r) {
            const int ly = y - r * step;
            if (ly < 0 || ly >= size) continue;
            const int top = r != 0 ? margin : 0, bot = r != n_rows - 1 ? margin : 0;
            if (ly < top || ly >= size - bot) continue;
            for (int q = c_lo; q <= c_hi; ++q) {
                const int lx = x - q * step;
                if (lx < 0 || lx >= size) continue;
                const int lef = q != 0 ? margin : 0, rig = q != n_cols - 1 ? margin : 0;
                if (lx < lef || lx >= size - rig) continue;
                sum += tiles[(((long)(r * n_cols + q) * c + ch) * size + ly) * size + lx];
                ++cnt;
            }
        }
        out[i] = cnt ? (uint8_t)(sum / cnt) : 0;
    }
}

static inline int grid1d(long total) { long b = (total + 255) / 256; return (int)(b < 16384 ? (b > 0 ? b : 1) : 16384); }

}  // namespace

extern "C" {

int pssr_sliding_tiles_u8(const uint8_t* sheet, float* tiles, int c, int h, int w, int size, int stride, int tile0, int ntile, pssr_stream_t s) {
    PSSR_CHECK(sheet && tiles && c > 0 && size > 0 && stride > 0 && h >= size && w >= size && ntile > 0 && tile0 >= 0, PSSR_ERR_ARG, "sliding_tiles: bad args");
    const int tiles_y = (h - size) / stride + 1, tiles_x = (w - size) / stride + 1;
    PSSR_CHECK(tile0 + ntile <= tiles_x * tiles_y, PSSR_ERR_ARG, "sliding_tiles: tile range %d+%d exceeds %d", tile0, ntile, tiles_x * tiles_y);
    hipLaunchKernelGGL(sliding_tiles_kernel, dim3(grid1d((long)ntile * c * size * size)), dim3(256), 0, (hipStream_t)s, sheet, tiles, c, h, w, size,
                       stride, tiles_x, tile0, ntile);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_patch_tiles_u8(const uint8_t* tiles, uint8_t* sheet, int c, int n_rows, int n_cols, int size, int overlap, int margin, pssr_stream_t s) {
    PSSR_CHECK(tiles && sheet && c > 0 && n_rows > 0 && n_cols > 0 && size > 0, PSSR_ERR_ARG, "patch_tiles: bad args");
    PSSR_CHECK(overlap >= 0 && overlap < size && margin >= 0 && margin <= overlap && 2 * margin < size, PSSR_ERR_ARG,
               "patch_tiles: need 0 <= margin <= overlap < size (margin=%d overlap=%d size=%d)", margin, overlap, size);
    const int step = size - overlap;
    const long total = (long)c * (n_rows * step + overlap) * (n_cols * step + overlap);
    hipLaunchKernelGGL(patch_tiles_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)s, tiles, sheet, c, n_rows, n_cols, size, overlap, margin);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

}  // extern "C"
