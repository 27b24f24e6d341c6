// Error reporting shared by the host translation units of liborbhip.so.
#ifndef ORBHIP_COMMON_H
#define ORBHIP_COMMON_H

// Records a printf-formatted message for orbhip_last_error() (per thread) and returns `code`.
int orb_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#endif
