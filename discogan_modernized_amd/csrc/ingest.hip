// Image ingest on the device: decoded uint8 rows -> the float NCHW batch the networks take.
// Replaces (reference file:line) the per-image host work of dataset.py:52-66 (read_images) / :239-254 (DiscoGANDataset):
//   domain 'A' (edges2*):  image[:, :256]; 255 - cv2.dilate(255 - image, ones(3,3))  == a 3x3 EROSION of the left half
//   domain 'B':            image[:, 256:]
//   cv2.resize(image, (S, S))  (INTER_LINEAR: half-pixel centres, edge clamp)
//   image.astype(np.float32) / 255. ; transpose(2, 0, 1)
// One launch per batch: a thread owns one output pixel (all 3 channels), reads its 2 x 2 source taps (each the minimum over
// the in-bounds 3 x 3 neighbourhood when `erode`), interpolates, normalises and writes one float into each channel plane
// (consecutive threads = consecutive x: coalesced stores; the uint8 reads hit L2).  The batch crosses PCIe as uint8
// (1 B / channel of the SOURCE size) instead of float32 of the output size.
//
// Arithmetic (mode):
//   0  float: coefficients in fp32, horizontal pass then vertical pass like cv2's generic path; the reference's domain 'A' image is
//      float64 at cv2.resize (255. - image), so its result is not rounded to uint8 -- neither is this one.
//   1  8-bit fixed point, cv2's CV_8U path: coefficients round(2048 a) as int16, rows S0 a0 + S1 a1 (int32), output
//      (((b0 (R0 >> 4)) >> 16) + ((b1 (R1 >> 4)) >> 16) + 2) >> 2, a uint8; then / 255 (correctly rounded fp32 division).
#include "dg_common.h"

struct PrepArgs {
    const uint8_t* src;
    float* dst;
    int N, H, W;        // source images [N][H][W][3]
    int x0, cw;         // crop: columns x0 .. x0 + cw - 1 (rows: all H)
    int erode, mode, S;
    double sx, sy;      // cw / S, H / S
};

__device__ __forceinline__ void prep_axis(int d, double scale, int n, int* i0, float* f) {
    // cv2 resize.cpp: fx = (dx + 0.5) * scale - 0.5; sx = floor(fx); fx -= sx; clamp at both ends with fx = 0
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(fx);
    fx -= (float)s;
    if (s < 0) { fx = 0.f; s = 0; }
    if (s >= n - 1) { fx = 0.f; s = n - 1; }
    *i0 = s;
    *f = fx;
}

__device__ __forceinline__ void prep_tap(const PrepArgs& p, const uint8_t* img, int y, int x, int v[3]) {
    // pixel (y, x) of the CROPPED image, eroded over its in-bounds 3 x 3 neighbourhood when p.erode
    if (!p.erode) {
        const uint8_t* q = img + ((long)y * p.W + p.x0 + x) * 3;
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
        return;
    }
    int m0 = 255, m1 = 255, m2 = 255;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)p.H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)p.cw) continue;        // the crop edge is an image border for cv2.dilate
            const uint8_t* q = img + ((long)yy * p.W + p.x0 + xx) * 3;
            m0 = min(m0, (int)q[0]); m1 = min(m1, (int)q[1]); m2 = min(m2, (int)q[2]);
        }
    }
    v[0] = m0; v[1] = m1; v[2] = m2;
}

__global__ __launch_bounds__(256) void image_prep_kernel(const PrepArgs p) {
    const long total = (long)p.N * p.S * p.S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % p.S);
        const int oy = (int)((i / p.S) % p.S);
        const long n = i / ((long)p.S * p.S);
        const uint8_t* img = p.src + n * (long)p.H * p.W * 3;
        int x0, y0;
        float fx, fy;
        prep_axis(ox, p.sx, p.cw, &x0, &fx);
        prep_axis(oy, p.sy, p.H, &y0, &fy);
        const int x1 = min(x0 + 1, p.cw - 1), y1 = min(y0 + 1, p.H - 1);
        int t00[3], t01[3], t10[3], t11[3];
        prep_tap(p, img, y0, x0, t00);
        prep_tap(p, img, y0, x1, t01);
        prep_tap(p, img, y1, x0, t10);
        prep_tap(p, img, y1, x1, t11);
        float out[3];
        if (p.mode == 1) {
            const int a1 = (int)rintf(fx * 2048.f), a0 = (int)rintf((1.f - fx) * 2048.f);
            const int b1 = (int)rintf(fy * 2048.f), b0 = (int)rintf((1.f - fy) * 2048.f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r0 = t00[c] * a0 + t01[c] * a1, r1 = t10[c] * a0 + t11[c] * a1;
                int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                v = min(max(v, 0), 255);
                out[c] = (float)v / 255.f;
            }
        } else {
            const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float r0 = (float)t00[c] * a0 + (float)t01[c] * a1, r1 = (float)t10[c] * a0 + (float)t11[c] * a1;
                out[c] = (r0 * b0 + r1 * b1) / 255.f;
            }
        }
        float* d = p.dst + (n * 3) * (long)p.S * p.S + (long)oy * p.S + ox;
#pragma unroll
        for (int c = 0; c < 3; ++c) d[(long)c * p.S * p.S] = out[c];
    }
}

extern "C" int dg_image_prep(const uint8_t* src, float* dst, int N, int H, int W, int x0, int cw, int erode, int mode, int S,
                             dg_stream_t s) {
    DG_CHECK_ARG(src && dst && N > 0 && H > 0 && W > 0 && S > 0, "dg_image_prep: bad argument");
    DG_CHECK_ARG(x0 >= 0 && cw > 0 && x0 + cw <= W, "dg_image_prep: crop [%d, %d) outside the %d-pixel rows", x0, x0 + cw, W);
    DG_CHECK_ARG(mode == 0 || mode == 1, "dg_image_prep: mode 0 (float) or 1 (cv2 8-bit fixed point)");
    PrepArgs p;
    p.src = src; p.dst = dst; p.N = N; p.H = H; p.W = W; p.x0 = x0; p.cw = cw; p.erode = erode ? 1 : 0; p.mode = mode; p.S = S;
    p.sx = (double)cw / S; p.sy = (double)H / S;
    const long total = (long)N * S * S;
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(image_prep_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, p);
    DG_CHECK_LAUNCH("image_prep");
    return DG_OK;
}
