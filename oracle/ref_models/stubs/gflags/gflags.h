// oracle/ref_models/stubs -- TEST INFRASTRUCTURE.  gflags is not installed in this image; core/common/Flags.h only needs the
// DEFINE_* / DECLARE_* macros to make plain globals named FLAGS_<name> (what gflags itself boils down to for a reader of the flag).
#pragma once
#include <cstdint>
#include <string>
#define DEFINE_int32(name, value, desc) int32_t FLAGS_##name = value
#define DEFINE_int64(name, value, desc) int64_t FLAGS_##name = value
#define DEFINE_bool(name, value, desc) bool FLAGS_##name = value
#define DEFINE_double(name, value, desc) double FLAGS_##name = value
#define DEFINE_string(name, value, desc) std::string FLAGS_##name = value
#define DECLARE_int32(name) extern int32_t FLAGS_##name
#define DECLARE_int64(name) extern int64_t FLAGS_##name
#define DECLARE_bool(name) extern bool FLAGS_##name
#define DECLARE_double(name) extern double FLAGS_##name
#define DECLARE_string(name) extern std::string FLAGS_##name
