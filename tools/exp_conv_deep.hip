// Development aid: times pcacc_conv3x3_deep_bf16 / pcacc_conv3x3_wgrad_deep_bf16 on one layer shape without torch, optionally with parts
// of the kernel compiled out (-DCD_EXP_...), to see what bounds it.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipcaccumulation_amd/csrc [-DCD_EXP_NOMFMA ...] tools/exp_conv_deep.hip -o /tmp/exp && /tmp/exp 20 72 72 128 128
#include "../pcaccumulation_amd/csrc/conv_deep.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 20, h = argc > 2 ? atoi(argv[2]) : 72, w = argc > 3 ? atoi(argv[3]) : 72;
    const int ci = argc > 4 ? atoi(argv[4]) : 128, co = argc > 5 ? atoi(argv[5]) : 128;
    const size_t nx = (size_t)n * h * w * ci, ny = (size_t)n * h * w * co, nw = (size_t)9 * co * ci;
    std::vector<uint16_t> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 25);
    for (size_t i = 0; i < nw; ++i) hw[i] = 0x3a00 + (uint16_t)((i * 40503u) >> 9 & 0x7f);
    uint16_t *dx, *dw, *dy;
    float *db, *gw, *gb;
    void *ws;
    hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&dy, ny * 2); hipMalloc(&db, co * 4);
    hipMalloc(&gw, nw * 4); hipMalloc(&gb, co * 4);
    hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemset(db, 0, co * 4);
    size_t wsb = 0;
    const bool wg = pcacc_conv3x3_wgrad_deep_workspace_bytes(n, h, w, ci, co, &wsb) == 0;
    hipMalloc(&ws, wsb ? wsb : 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 30;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 3; ++i) {
            if (pass == 0) pcacc_conv3x3_deep_bf16(dx, nullptr, dw, db, dy, n, h, w, ci, co, 1, nullptr);
            else if (wg) pcacc_conv3x3_wgrad_deep_bf16(dy, nullptr, dx, gw, gb, n, h, w, ci, co, ws, wsb, nullptr);
        }
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) {
            if (pass == 0) pcacc_conv3x3_deep_bf16(dx, nullptr, dw, db, dy, n, h, w, ci, co, 1, nullptr);
            else if (wg) pcacc_conv3x3_wgrad_deep_bf16(dy, nullptr, dx, gw, gb, n, h, w, ci, co, ws, wsb, nullptr);
        }
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters, fl = 2.0 * n * h * w * co * ci * 9;
        printf("%s n=%d %dx%d %d->%d : %.1f us  %.0f TFLOP/s  err=%d\n", pass ? "wgrad" : "fwd  ", n, h, w, ci, co, us, fl / us / 1e6,
               (int)hipGetLastError());
    }
    return 0;
}
