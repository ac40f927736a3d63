// pt_internal.hpp -- shared between the translation units of libptamd.so.
#pragma once

#include <string>

namespace ptint {

// Record `msg` as the calling thread's last error and return `code`.
int fail(int code, const std::string& msg);

// For the handler of a function-try-block around the body of an extern "C" entry point: turns the exception in
// flight into a status (out of host memory -> PT_ERR_LIMIT), so that none crosses the C boundary.
int caught();

} // namespace ptint
