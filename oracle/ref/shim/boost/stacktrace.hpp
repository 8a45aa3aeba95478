// Oracle-build-only shim: boost is not installed; the reference only prints a
// stack trace when it aborts (sxt/base/error/stacktrace.cc).
#pragma once
#include <string>
namespace boost::stacktrace {
struct stacktrace {};
inline std::string to_string(const stacktrace&) { return "<stacktrace unavailable in oracle build>"; }
} // namespace boost::stacktrace
