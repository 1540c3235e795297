#include "common.h"
static thread_local std::string g_err;
void ydl_set_error(const std::string& s) { g_err = s; }
extern "C" const char* ydl_last_error(void) { return g_err.c_str(); }
extern "C" int ydl_version(void) { return 1; }
