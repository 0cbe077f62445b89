// Are fp64 sqrt, divide and hypot on gfx950 bit-identical to the host's (IEEE sqrtsd/divsd, glibc hypot)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
__global__ void k(const double *a, const double *b, double *s, double *d, double *h, double *f, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = sqrt(a[i]);
    d[i] = a[i] / b[i];
    h[i] = hypot(a[i], b[i]);
    f[i] = a[i] * b[i] + a[i]; // contraction probe
}
int main()
{
    const int n = 1 << 20;
    std::mt19937_64 rng(1);
    std::vector<double> a(n), b(n), s(n), d(n), h(n), f(n);
    std::uniform_real_distribution<double> U(-20, 20);
    for (int i = 0; i < n; i++) { a[i] = std::fabs(std::ldexp(1.0 + (rng() >> 12) * 0x1p-52, (int)U(rng))); b[i] = std::ldexp(1.0 + (rng() >> 12) * 0x1p-52, (int)U(rng)); }
    double *da, *db, *ds, *dd, *dh, *df;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dd, n * 8); hipMalloc(&dh, n * 8); hipMalloc(&df, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, ds, dd, dh, df, n);
    hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(h.data(), dh, n * 8, hipMemcpyDeviceToHost); hipMemcpy(f.data(), df, n * 8, hipMemcpyDeviceToHost);
    long ms = 0, md = 0, mh = 0, mf = 0;
    for (int i = 0; i < n; i++) {
        volatile double p = a[i] * b[i];
        double hs = std::sqrt(a[i]), hd = a[i] / b[i], hh = std::hypot(a[i], b[i]), hf = p + a[i];
        ms += memcmp(&hs, &s[i], 8) != 0; md += memcmp(&hd, &d[i], 8) != 0; mh += memcmp(&hh, &h[i], 8) != 0; mf += memcmp(&hf, &f[i], 8) != 0;
    }
    printf("mismatches of %d: sqrt %ld, div %ld, hypot %ld, mul+add(no fma) %ld\n", n, ms, md, mh, mf);
    return 0;
}
