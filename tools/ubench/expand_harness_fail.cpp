#include <cstdarg>
#include <cstdio>
#include <string>
static thread_local std::string g;
int mhx_fail(int code, const char* fmt, ...) { char b[1024]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); g = b; fprintf(stderr, "%s\n", b); return code; }
extern "C" const char* mhx_last_error(void) { return g.c_str(); }
