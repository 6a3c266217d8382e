// oracle/ref_processor/stubs -- TEST INFRASTRUCTURE.  boost/regex.hpp for the build of the REFERENCE's ProcessorParseRegexNative.cpp in
// an image without boost: the types and the one call that file makes (through common/StringTools.h BoostRegexMatch), answered by the
// oracle's matcher (oracle/bt_regex.c, the restatement of boost::regex_match this repo pins on third-party vectors).  The processor code
// around the match -- Init, Process, ProcessEvent, RegexLogLineParser, AddLog -- is the reference's own, compiled from where it lies.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "bt_regex.h"  // oracle/

// (boost/config: ProcessorFilterNative.cpp uses the branch hints)
#ifndef BOOST_LIKELY
#define BOOST_LIKELY(x) __builtin_expect(!!(x), 1)
#define BOOST_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif

namespace boost {
typedef unsigned match_flag_type;
constexpr match_flag_type match_default = 0, match_continuous = 1;
struct regex_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
class regex {
public:
    enum flag_type_ { normal = 0, save_subexpression_location = 1 };
    regex() = default;
    explicit regex(const std::string& p, unsigned = 0) : mText(p) {
        char err[256] = {0};
        orx_prog* prog = orx_compile(p.data(), p.size(), 0, err, sizeof err);
        if (!prog) throw regex_error(err);
        mProg.reset(prog, orx_free);
    }
    unsigned mark_count() const { return mProg ? unsigned(orx_mark_count(mProg.get())) : 0u; }
    const std::string& str() const { return mText; }
    const orx_prog* prog() const { return mProg.get(); }

private:
    std::string mText;
    std::shared_ptr<orx_prog> mProg;
};
template <class It>
struct sub_match {
    It first{}, second{};
    bool matched = false;
    It begin() const { return first; }
    It end() const { return second; }
    std::ptrdiff_t length() const { return matched ? second - first : 0; }
    std::string str() const { return matched ? std::string(first, second) : std::string(); }
};
template <class It>
class match_results {
public:
    size_t size() const { return mSubs.size(); }
    bool empty() const { return mSubs.empty(); }
    const sub_match<It>& operator[](size_t i) const { return mSubs[i]; }
    std::vector<sub_match<It>> mSubs;
};
typedef match_results<const char*> cmatch;
// whole-sequence match with sub-matches: boost::regex_match(first, last, what, re, flags).  An unmatched group is {last, last},
// matched = false (boost's sub_match of a group that did not take part); the complexity exception is a std::runtime_error there too.
inline bool regex_match(const char* first, const char* last, match_results<const char*>& what, const regex& re,
                        match_flag_type = match_default) {
    what.mSubs.clear();
    if (!re.prog()) return false;
    const int groups = orx_mark_count(re.prog());
    std::vector<int32_t> caps(size_t(2) * size_t(groups + 1), -1);
    const int r = orx_fullmatch(re.prog(), reinterpret_cast<const uint8_t*>(first), size_t(last - first), caps.data());
    if (r < 0) throw std::runtime_error("The complexity of matching the regular expression exceeded predefined bounds.");
    if (r == 0) return false;
    what.mSubs.resize(size_t(groups + 1));
    for (int g = 0; g <= groups; ++g) {
        sub_match<const char*>& s = what.mSubs[size_t(g)];
        if (caps[size_t(2 * g)] >= 0) {
            s.first = first + caps[size_t(2 * g)];
            s.second = first + caps[size_t(2 * g + 1)];
            s.matched = true;
        } else {
            s.first = s.second = last;
        }
    }
    return true;
}
// whole-sequence match without sub-matches
inline bool regex_match(const char* first, const char* last, const regex& re, match_flag_type flags = match_default) {
    match_results<const char*> what;
    return regex_match(first, last, what, re, flags);
}
// boost::regex_search(first, last, what, re, match_continuous): the match must begin at `first` (what the multiline processor asks);
// without the flag: leftmost match anywhere
inline bool regex_search(const char* first, const char* last, match_results<const char*>& what, const regex& re,
                         match_flag_type flags = match_default) {
    what.mSubs.clear();
    if (!re.prog()) return false;
    const int groups = orx_mark_count(re.prog());
    std::vector<int32_t> caps(size_t(2) * size_t(groups + 1), -1);
    const uint8_t* s = reinterpret_cast<const uint8_t*>(first);
    const int r = (flags & match_continuous) ? orx_prefixmatch(re.prog(), s, size_t(last - first), caps.data())
                                             : orx_search(re.prog(), s, size_t(last - first), 0, caps.data());
    if (r < 0) throw std::runtime_error("The complexity of matching the regular expression exceeded predefined bounds.");
    if (r == 0) return false;
    what.mSubs.resize(size_t(groups + 1));
    for (int g = 0; g <= groups; ++g) {
        sub_match<const char*>& m = what.mSubs[size_t(g)];
        if (caps[size_t(2 * g)] >= 0) {
            m.first = first + caps[size_t(2 * g)];
            m.second = first + caps[size_t(2 * g + 1)];
            m.matched = true;
        } else {
            m.first = m.second = last;
        }
    }
    return true;
}
}  // namespace boost
