// Host-side helpers shared by the C-ABI translation units: thread-local error text, launch check.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

namespace vk {
char* error_buffer();                       // thread-local, 512 bytes
int set_error(const char* fmt, ...);        // formats into the buffer, returns -1
int check_launch(const char* what);         // hipGetLastError() -> 0 / -1
}  // namespace vk
