// Argument streams and checksum terms shared by the device self-test (k_selftest) and its host
// twin (oracle/mathcheck.cpp): function f evaluated at pseudo-random argument i contributes
// bits(result) * (2i+1) to a 64-bit sum.  Equal sums on host and device mean the device
// evaluates the restated library routines to the same bits as the host build, which
// tests/test_math_host.py has compared with torch.
#pragma once
#include "tclip_math.h"

namespace tclip {

constexpr uint32_t kSelfTestCount = 1u << 22;
constexpr int kSelfTestFunctions = 8;

TCLIP_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// positive float with exponent drawn from [lo_exp, lo_exp + n_exp) and a random mantissa
TCLIP_HD float rand_float(uint32_t i, uint32_t salt, int lo_exp, uint32_t n_exp) {
    const uint32_t h1 = mix32(i * 2654435761u + salt), h2 = mix32(i ^ (salt * 0x9e3779b9u));
    return bits_f32(((uint32_t)(127 + lo_exp) + (h2 % n_exp)) << 23 | (h1 & 0x7fffffu));
}

TCLIP_HD unsigned long long selftest_term(int f, uint32_t i, const LogTabEntry* tab) {
    float r;
    switch (f) {
        case 0: r = digamma_f32(rand_float(i, 11u, -20, 50)); break;                 // 2^-20 .. 2^30
        case 1: r = lgamma_f32(rand_float(i, 12u, -30, 60)); break;                  // 2^-30 .. 2^30
        case 2: r = sqrt_torch_f32(rand_float(i, 13u, -40, 80)); break;
        case 3: r = exp_f32_sleef(-rand_float(i, 14u, -10, 17)); break;              // -2^-10 .. -2^7
        case 4: r = log_f32(rand_float(i, 15u, -50, 51)); break;
        case 5: { float p, l; digamma_lgamma_xp1(rand_float(i, 16u, -45, 70), tab, p, l); r = p; break; }
        case 6: { float p, l; digamma_lgamma_xp1(rand_float(i, 16u, -45, 70), tab, p, l); r = l; break; }
        default: r = digamma_pos_f32(rand_float(i, 17u, -20, 50), tab); break;
    }
    return (unsigned long long)f32_bits(r) * (unsigned long long)(2u * i + 1u);
}

}  // namespace tclip
