// Detect head inference decode (reference models/head/yolov5_head.py:68-78, _make_grid_old :127-136).
//   y = sigmoid(raw);  xy = (2y - 0.5 + grid) * stride;  wh = (2y)^2 * (anchors*stride);  rest = y
// Reads the head conv output in place through element strides (the NHWC GEMM output is viewed as
// (B, na, ny, nx, no); no permute/contiguous copy is materialised) and writes this level's slice of
// z (B, A_total, no) fp32.  HBM-bound elementwise kernel: one workgroup per (b, a, y) output row,
// lanes run over the contiguous x*no + c axis.
#include "et_device.h"
#include "../../include/et_hip.h"

template <typename T>
__global__ __launch_bounds__(256) void detect_decode_kernel(const T* __restrict__ raw, int na, int ny, int nx, int no,
                                                            long long sb, long long sa, long long sy, long long sx,
                                                            const float* __restrict__ anchor_px, float stride,
                                                            float* __restrict__ z, long long A_total, long long a_off) {
    const int row = blockIdx.x;               // (b*na + a)*ny + y
    const int y = row % ny;
    const int a = (row / ny) % na;
    const int b = row / (ny * na);
    const T* src = raw + b * sb + a * sa + y * sy;
    float* dst = z + ((long long)b * A_total + a_off + ((long long)a * ny + y) * nx) * no;
    const float aw = anchor_px[a * 2 + 0], ah = anchor_px[a * 2 + 1];
    const int n = nx * no;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int x = t / no, c = t - x * no;
        const float v = et_elem<T>::ld(src[x * sx + c]);
        float s = 1.0f / (1.0f + expf(-v));
        if (c < 2) {
            const float g = c == 0 ? (float)x : (float)y;
            s = (s * 2.0f - 0.5f + g) * stride;
        } else if (c < 4) {
            const float t2 = s * 2.0f;
            s = (t2 * t2) * (c == 2 ? aw : ah);
        }
        dst[t] = s;
    }
}

extern "C" int et_detect_decode(const void* raw, int dtype, int B, int na, int ny, int nx, int no,
                                int64_t sb, int64_t sa, int64_t sy, int64_t sx, const float* anchor_px,
                                float stride, float* z, int64_t A_total, int64_t a_offset, et_stream_t stream) {
    if (!raw || !anchor_px || !z) return -1;
    if (B <= 0 || na <= 0 || ny <= 0 || nx <= 0 || no < 5) return -2;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(B * na * ny), block(256);
    if (dtype == ET_F32)
        hipLaunchKernelGGL((detect_decode_kernel<float>), grid, block, 0, s, (const float*)raw, na, ny, nx, no,
                           (long long)sb, (long long)sa, (long long)sy, (long long)sx, anchor_px, stride, z,
                           (long long)A_total, (long long)a_offset);
    else if (dtype == ET_BF16)
        hipLaunchKernelGGL((detect_decode_kernel<uint16_t>), grid, block, 0, s, (const uint16_t*)raw, na, ny, nx, no,
                           (long long)sb, (long long)sa, (long long)sy, (long long)sx, anchor_px, stride, z,
                           (long long)A_total, (long long)a_offset);
    else if (dtype == ET_F16)
        hipLaunchKernelGGL((detect_decode_kernel<et_f16>), grid, block, 0, s, (const et_f16*)raw, na, ny, nx, no,
                           (long long)sb, (long long)sa, (long long)sy, (long long)sx, anchor_px, stride, z,
                           (long long)A_total, (long long)a_offset);
    else
        return -2;
    ET_CHECK_LAUNCH();
    return 0;
}
