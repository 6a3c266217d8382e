// oracle/ref_models/stubs -- TEST INFRASTRUCTURE.  The two content keys CommonParserOptions::ShouldEraseEvent names
// (core/plugin/processor/inner/ProcessorParseContainerLogNative.h:50-51); their VALUES are taken from the reference's .cpp at build time
// (oracle/ref_models/Makefile, link_stubs.cpp).
#pragma once
#include <string>

namespace logtail {
class ProcessorParseContainerLogNative {
public:
    static const std::string containerTimeKey;
    static const std::string containerSourceKey;
};
}  // namespace logtail
