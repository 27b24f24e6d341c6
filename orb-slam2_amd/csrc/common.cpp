#include "common.h"

#include <cstdarg>
#include <cstdio>

#include "../../include/orbhip.h"

static thread_local char g_last_error[512] = "";

int orb_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" {

const char* orbhip_last_error(void) { return g_last_error; }
const char* orbhip_version(void) { return "orbhip 0.1 (gfx950)"; }

// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1805-1821): the reference's SWAR bit count
// over eight 32-bit words equals the population count of the xor.
int orbhip_hamming_distance(const void* a, const void* b) {
    const unsigned long long* pa = (const unsigned long long*)a;
    const unsigned long long* pb = (const unsigned long long*)b;
    unsigned long long wa[4], wb[4];
    __builtin_memcpy(wa, pa, 32);
    __builtin_memcpy(wb, pb, 32);
    return __builtin_popcountll(wa[0] ^ wb[0]) + __builtin_popcountll(wa[1] ^ wb[1]) +
           __builtin_popcountll(wa[2] ^ wb[2]) + __builtin_popcountll(wa[3] ^ wb[3]);
}

}  // extern "C"
