// Library-level entry points: version and thread-local error text.
#include "be_common.h"

namespace be {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace be

extern "C" int be_version(void) { return 1; }
extern "C" const char* be_last_error(void) { return be::last_error_buf(); }
