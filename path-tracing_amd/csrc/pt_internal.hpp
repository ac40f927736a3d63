// pt_internal.hpp -- shared between the translation units of libptamd.so.
#pragma once

#include <string>

namespace ptint {

// Record `msg` as the calling thread's last error and return `code`.
int fail(int code, const std::string& msg);

} // namespace ptint
