// abi_check.cpp -- a C++ host built against include/gsplat.h and libgsplat_hip.so only (no Python, no torch):
// checks the ABI version, that the header-only mirror compiles, and that creating a context without a usable GPU
// fails loudly with GS_ERR_NO_DEVICE instead of falling back to anything.
//   g++ -std=c++17 -I. host/abi_check.cpp -Lgaussiansplattingmlx_amd -lgsplat_hip -Wl,-rpath,$PWD/gaussiansplattingmlx_amd
#include <cstdio>

#include "GaussianRenderer.hpp"

int main()
{
    if (gs_abi_version() != GSPLAT_ABI_VERSION) { std::printf("ABI version mismatch\n"); return 2; }
    float win[121];
    if (gs_ssim_window(11, 1.5f, win) != GS_OK) return 3;          // host-only entry point
    double sum = 0;
    for (float w : win) sum += w;
    try {
        gsplat::GaussianRenderer r(4, 64, 64, gsplat::TILE_SIZE_H_W{16, 16}, false);
        std::printf("context created on a GPU; window sum %.6f\n", sum);
        return 0;
    } catch (const gsplat::Error& e) {
        std::printf("no context: code %d (%s); window sum %.6f\n", e.code, e.what(), sum);
        return e.code == GS_ERR_NO_DEVICE ? 10 : 4;
    }
}
