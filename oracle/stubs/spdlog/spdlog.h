// Build-only stand-in for <spdlog/spdlog.h> (+ the bundled fmt it drags in). TEST INFRASTRUCTURE for
// oracle/Makefile: every log call in the reference sources becomes a no-op, fmt::format returns "".
#pragma once
#include <cstring>
#include <memory>
#include <string>

namespace fmt {
template <typename... Args>
inline std::string format(const char*, const Args&...) { return std::string(); }
template <typename... Args>
inline std::string format(const std::string&, const Args&...) { return std::string(); }
}  // namespace fmt

namespace spdlog {
namespace level {
enum level_enum { trace = 0, debug = 1, info = 2, warn = 3, err = 4, critical = 5, off = 6 };
}
class logger {
 public:
  template <typename... Args> void trace(const char*, const Args&...) {}
  template <typename... Args> void debug(const char*, const Args&...) {}
  template <typename... Args> void info(const char*, const Args&...) {}
  template <typename... Args> void warn(const char*, const Args&...) {}
  template <typename... Args> void error(const char*, const Args&...) {}
  template <typename... Args> void critical(const char*, const Args&...) {}
  void flush() {}
};
}  // namespace spdlog
