// oracle/ref_models/stubs -- TEST INFRASTRUCTURE.  Stands in for core/common/ParamExtractor.h: the reference's CommonParserOptions::Init
// reads its four options through these; with the inert Json::Value of tests/refhdr nothing can be read, so Init is compiled but never
// called -- the tests set the (public) option fields and call the three policy functions, which are the reference's own code.
#pragma once
#include <string>

#include "json/json.h"

namespace logtail {
inline bool GetOptionalBoolParam(const Json::Value&, const std::string&, bool&, std::string&) { return true; }
inline bool GetOptionalStringParam(const Json::Value&, const std::string&, std::string&, std::string&) { return true; }
}  // namespace logtail
#define PARAM_WARNING_IGNORE(...) \
    do {                          \
    } while (0)
#define PARAM_WARNING_DEFAULT(...) \
    do {                           \
    } while (0)
