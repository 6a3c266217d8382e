// oracle/ref_processor/stubs -- TEST INFRASTRUCTURE.  common/StringTools.h includes it; nothing compiled here calls it.
#pragma once
#include <sstream>
namespace boost {
template <class T, class S>
T lexical_cast(const S& s) {
    std::stringstream ss;
    ss << s;
    T t{};
    ss >> t;
    return t;
}
}  // namespace boost
