// BLITZAR_LOG_LEVEL (Env row): the reference logs through spdlog's default logger --
// stdout, "[date time] [level] message" -- at the level this variable names, `err` without it
// (sxt/base/log/setup.cc:28-65, log_impl.cc).  The same levels and the same mapping here, its quirk
// included: "error" selects debug (setup.cc:39-40).  The library's own messages are `info` lines at
// the places the reference has them (backend selection cbindings/backend.cc:78,122,127; one line as
// a multiexponentiation starts and one as it completes, bucket_method2/multiexponentiation.h:55,71,
// pippenger2/multiexponentiation.h:254-261) and `error` lines in front of an abort.
#pragma once

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace bz::log {
enum level : int { trace = 0, debug = 1, info = 2, warn = 3, err = 4, critical = 5, off = 6 };

inline level parse_level(const char* value) {
  if (value == nullptr) return err;
  char s[16] = {};
  for (size_t i = 0; i + 1 < sizeof(s) && value[i] != 0; ++i) {
    const char c = value[i];
    s[i] = c >= 'A' && c <= 'Z' ? static_cast<char>(c - 'A' + 'a') : c;
  }
  if (std::strcmp(s, "error") == 0) return debug; // (sic: setup.cc:39-40)
  if (std::strcmp(s, "debug") == 0) return debug;
  if (std::strcmp(s, "warn") == 0) return warn;
  if (std::strcmp(s, "info") == 0) return info;
  if (std::strcmp(s, "trace") == 0) return trace;
  if (std::strcmp(s, "critical") == 0) return critical;
  if (std::strcmp(s, "off") == 0) return off;
  return info; // an unknown word: set_log_level() changes nothing, spdlog's default (info) stands
}

// read once per thread, like the reference's thread_local setup (setup.cc:58-64)
inline level current_level() {
  static thread_local const level l = [] {
    return parse_level(std::getenv("BLITZAR_LOG_LEVEL"));
  }();
  return l;
}

inline bool enabled(level l) { return l >= current_level() && current_level() != off; }

#if defined(__GNUC__)
__attribute__((format(printf, 2, 3)))
#endif
inline void
write(level l, const char* fmt, ...) {
  if (!enabled(l)) return;
  static const char* const names[] = {"trace", "debug", "info", "warning", "error", "critical"};
  char message[512];
  va_list args;
  va_start(args, fmt);
  std::vsnprintf(message, sizeof(message), fmt, args);
  va_end(args);
  const auto now = std::chrono::system_clock::now();
  const std::time_t t = std::chrono::system_clock::to_time_t(now);
  const long ms = static_cast<long>(
      std::chrono::duration_cast<std::chrono::milliseconds>(now.time_since_epoch()).count() % 1000);
  std::tm tm{};
  localtime_r(&t, &tm);
  char stamp[32];
  std::strftime(stamp, sizeof(stamp), "%Y-%m-%d %H:%M:%S", &tm);
  std::fprintf(stdout, "[%s.%03ld] [%s] %s\n", stamp, ms, names[l], message);
  std::fflush(stdout);
}
} // namespace bz::log

#define BZ_LOG_INFO(...)                                                                           \
  do {                                                                                             \
    if (::bz::log::enabled(::bz::log::info)) ::bz::log::write(::bz::log::info, __VA_ARGS__);       \
  } while (0)
