// version.hip -- library identity + per-thread error string of the C ABI.
#include "common.h"
extern "C" int segx_version(void) { return 100; }
extern "C" int segx_last_error(char* buf, int buflen) {
    const char* e = segx::err_buf();
    int n = (int)strlen(e);
    if (buf && buflen > 0) { int c = n < buflen - 1 ? n : buflen - 1; memcpy(buf, e, c); buf[c] = 0; }
    return n;
}
