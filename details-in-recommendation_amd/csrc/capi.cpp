// capi.cpp -- version / error plumbing of libdir_hip.so.
#include "common.hpp"

namespace dir {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace dir

extern "C" int dir_version(void) { return DIR_VERSION; }
extern "C" const char* dir_last_error(void) { return dir::err_buf(); }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slice-by-8: the checksum of TensorFlow's checkpoint bundles
// (checkpoint.py / tf_bundle.py write and verify it over GB-sized tables; a pure-Python loop manages ~10 MB/s).  HOST function.
namespace {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
};
}  // namespace

extern "C" uint32_t dir_crc32c(uint32_t crc, const void* data, int64_t n) {
    static const Crc32cTables T;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xffffffffu;
    while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
        --n;
    }
    while (n >= 8) {
        uint64_t v;
        __builtin_memcpy(&v, p, 8);
        v ^= c;
        c = T.t[7][v & 0xff] ^ T.t[6][(v >> 8) & 0xff] ^ T.t[5][(v >> 16) & 0xff] ^ T.t[4][(v >> 24) & 0xff] ^
            T.t[3][(v >> 32) & 0xff] ^ T.t[2][(v >> 40) & 0xff] ^ T.t[1][(v >> 48) & 0xff] ^ T.t[0][(v >> 56) & 0xff];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}
