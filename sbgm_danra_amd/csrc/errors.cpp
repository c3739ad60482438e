#include <cstdarg>
#include <cstdio>
static thread_local char g_err[1024] = "";
void sbgm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* sbgm_get_error() { return g_err; }
