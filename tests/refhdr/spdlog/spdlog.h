// Declarations-only stand-in for spdlog (not in this image): exactly what core/logger/Logger.h touches, so that the shim can be
// type-checked against the reference's CollectionPipelineContext / AlarmManager (tests/test_refhdr_compile.py).  Never linked.
#pragma once
#include <memory>
#include <string>
namespace spdlog {
namespace level {
enum level_enum { trace = 0, debug = 1, info = 2, warn = 3, err = 4, critical = 5, off = 6 };
}
class logger {
public:
    bool should_log(level::level_enum) const;
    template <typename... Args>
    void log(level::level_enum, const char*, const Args&...) {}  // (a body: oracle/ref_models links the reference's model sources)
    void log(level::level_enum, const std::string&);
    void flush();
};
}  // namespace spdlog
