// tests/refhdr -- TEST INFRASTRUCTURE: a declarations-only stand-in for <boost/utility/string_view.hpp> (Boost is not installed in
// this image), just enough for the reference's own headers (core/common/StringView.h, core/models/*.h) to be parsed, so that the
// shim's LC_USE_REFERENCE_HEADERS variant can be type-checked against the REAL LogEvent / PipelineEventGroup declarations
// (tests/test_refhdr_compile.py).  boost::string_view has std::string_view's interface (plus to_string / clear).  Header-only: the type-check never runs it; oracle/ref_models
// (the reference's own event model compiled for the tests) does.
#pragma once
#include <algorithm>  // (the real header pulls these in; the reference headers lean on that)
#include <cstddef>
#include <cstring>
#include <ostream>
#include <string>
#include <string_view>

namespace boost {
class string_view : public std::string_view {
public:
    using std::string_view::string_view;
    constexpr string_view() noexcept = default;
    constexpr string_view(std::string_view s) noexcept : std::string_view(s) {}
    string_view(const std::string& s) noexcept : std::string_view(s) {}
    std::string to_string() const { return std::string(data(), size()); }
    void clear() noexcept { *this = string_view(); }
    constexpr string_view substr(size_type pos = 0, size_type n = npos) const { return string_view(std::string_view::substr(pos, n)); }
    bool starts_with(string_view x) const noexcept { return size() >= x.size() && compare(0, x.size(), x) == 0; }
    bool ends_with(string_view x) const noexcept { return size() >= x.size() && compare(size() - x.size(), npos, x) == 0; }
};
}  // namespace boost

namespace std {
template <>
struct hash<boost::string_view> {
    size_t operator()(const boost::string_view& s) const noexcept { return hash<std::string_view>()(s); }
};
}  // namespace std
