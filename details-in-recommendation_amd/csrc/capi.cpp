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
