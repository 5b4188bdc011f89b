// metrics.hip -- the reference's objective metrics core on the device: PSNR, SSIM
// (11x11 Gaussian, sigma 1.5, reflect-101 borders) and dE-ITP (BT.2124) between two unit-range images
// [3][H][W] (R, G, B planes; f16 or f32) -- src/gui_objective_metrics.py:438-528 as restated in
// oracle/metrics_oracle.py (parity unpinned: that module needs cv2).  One 16x16-pixel tile per workgroup,
// per-workgroup fp64 partial sums, summed in index order on the host: results are reproducible bit for bit.
#include "launchers.h"

namespace {

constexpr int MT = 16, R = 5, HT = MT + 2 * R;   // 26 x 26 halo tile
__constant__ float c_gauss[11];

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

template <typename T>
__device__ __forceinline__ float ld(const void *p, size_t i) { return (float)reinterpret_cast<const T *>(p)[i]; }

__device__ __forceinline__ float pq_oetf(float lum)
{
    const float y = fminf(fmaxf(lum / 10000.0f, 0.f), 1.f);
    const float ym = powf(y, 2610.0f / 16384.0f);
    const float num = 3424.0f / 4096.0f + (2413.0f / 128.0f) * ym;
    const float den = 1.0f + (2392.0f / 128.0f) * ym;
    return powf(num / fmaxf(den, 1e-12f), 2523.0f / 32.0f);
}

__device__ __forceinline__ void to_itp(const float rgb[3], float out[3])
{
    const float l = (1688.0f * rgb[0] + 2146.0f * rgb[1] + 262.0f * rgb[2]) / 4096.0f;
    const float m = (683.0f * rgb[0] + 2951.0f * rgb[1] + 462.0f * rgb[2]) / 4096.0f;
    const float s = (99.0f * rgb[0] + 309.0f * rgb[1] + 3688.0f * rgb[2]) / 4096.0f;
    const float lp = pq_oetf(l), mp = pq_oetf(m), sp = pq_oetf(s);
    out[0] = 0.5f * lp + 0.5f * mp;
    out[1] = 0.5f * ((6610.0f * lp - 13613.0f * mp + 7003.0f * sp) / 4096.0f);
    out[2] = (17933.0f * lp - 17390.0f * mp - 543.0f * sp) / 4096.0f;
}

template <typename T>
__global__ __launch_bounds__(256) void metrics_kernel(MetricsParams p)
{
    __shared__ float s_a[HT][HT + 1], s_b[HT][HT + 1];
    __shared__ float s_h[5][HT][MT + 1];          // horizontally blurred a, b, a*a, b*b, a*b
    __shared__ double s_red[3][256];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * MT, y0 = blockIdx.y * MT;
    const int x = x0 + tx, y = y0 + ty;
    const bool in = x < p.W && y < p.H;
    const size_t plane = (size_t)p.H * p.W;
    double se = 0.0, ss = 0.0, de = 0.0;
    float ca[3], cb[3];
    for (int c = 0; c < 3; ++c) {
        __syncthreads();
        for (int e = tid; e < HT * HT; e += 256) {
            const int r = e / HT, q = e % HT;
            const size_t idx = c * plane + (size_t)reflect101(y0 - R + r, p.H) * p.W + reflect101(x0 - R + q, p.W);
            s_a[r][q] = ld<T>(p.a, idx);
            s_b[r][q] = ld<T>(p.b, idx);
        }
        __syncthreads();
        for (int e = tid; e < HT * MT; e += 256) {
            const int r = e / MT, q = e % MT;
            float h[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < 11; ++i) {
                const float a = s_a[r][q + i], b = s_b[r][q + i], k = c_gauss[i];
                h[0] += a * k; h[1] += b * k; h[2] += (a * a) * k; h[3] += (b * b) * k; h[4] += (a * b) * k;
            }
            for (int j = 0; j < 5; ++j) s_h[j][r][q] = h[j];
        }
        __syncthreads();
        ca[c] = s_a[ty + R][tx + R];
        cb[c] = s_b[ty + R][tx + R];
        if (in) {
            float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < 11; ++i)
                for (int j = 0; j < 5; ++j) v[j] += s_h[j][ty + i][tx] * c_gauss[i];
            const float mu_a = v[0], mu_b = v[1];
            const float sa = v[2] - mu_a * mu_a, sb = v[3] - mu_b * mu_b, sab = v[4] - mu_a * mu_b;
            const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
            const float num = (2.0f * mu_a * mu_b + c1) * (2.0f * sab + c2);
            const float den = (mu_a * mu_a + mu_b * mu_b + c1) * (sa + sb + c2);
            ss += (double)(num / (den + 1e-12f));
            const float d = ca[c] - cb[c];
            se += (double)(d * d);
        }
    }
    if (in) {
        float ra[3], rb[3], ia[3], ib[3];
        for (int c = 0; c < 3; ++c) {
            ra[c] = fminf(fmaxf(ca[c], 0.f), 1.f) * p.peak_nits;
            rb[c] = fminf(fmaxf(cb[c], 0.f), 1.f) * p.peak_nits;
        }
        to_itp(ra, ia);
        to_itp(rb, ib);
        const float di = ia[0] - ib[0], dt = ia[1] - ib[1], dp = ia[2] - ib[2];
        de = (double)(720.0f * sqrtf(di * di + dt * dt + dp * dp + 1e-12f));
    }
    s_red[0][tid] = se; s_red[1][tid] = ss; s_red[2][tid] = de;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {          // fixed tree: the same sum every run
        if (tid < o)
            for (int j = 0; j < 3; ++j) s_red[j][tid] += s_red[j][tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        double *out = p.partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        out[0] = s_red[0][0]; out[1] = s_red[1][0]; out[2] = s_red[2][0];
    }
}

}  // namespace

int metrics_blocks(int H, int W) { return ((W + MT - 1) / MT) * ((H + MT - 1) / MT); }

hipError_t metrics_launch(const MetricsParams &p, hipStream_t s)
{
    static bool init = false;
    if (!init) {
        float k[11];
        double sum = 0.0, kd[11];
        for (int i = 0; i < 11; ++i) { const double x = i - 5.0; kd[i] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); sum += kd[i]; }
        for (int i = 0; i < 11; ++i) k[i] = (float)(kd[i] / sum);
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), k, sizeof k);
        if (e != hipSuccess) return e;
        init = true;
    }
    const dim3 grid((p.W + MT - 1) / MT, (p.H + MT - 1) / MT);
    if (p.is_f32) hipLaunchKernelGGL(metrics_kernel<float>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(metrics_kernel<f16>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}
