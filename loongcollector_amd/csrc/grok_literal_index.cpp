// grok_literal_index.cpp -- host half of the Grok matcher's literal index (grok_kernel.hpp grok_literal_index_kernel): the
// required literals of a whole Match list as ONE Aho-Corasick automaton, completed into a DFA over byte classes.
// Blob layout: grok_literal_layout.h.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "grok_literal_layout.h"
#include "grok_runtime.hpp"
#include "regex_handle.hpp"

constexpr size_t kGrokLiteralBytes = 32;  // a longer literal is represented by its last 32 bytes (as the per-entry filter did)
std::string lcGrokLiteralOf(const lc_regex* re) {
    const std::string& s = re->requiredLiteral;
    return s.size() > kGrokLiteralBytes ? s.substr(s.size() - kGrokLiteralBytes) : s;
}
std::vector<uint32_t> lcBuildGrokLiteralBlob(const std::vector<std::string>& lits) {
    // byte classes: every byte that occurs in a literal is its own class, class 0 = everything else
    std::vector<uint8_t> cmap(256, 0);
    uint32_t ncls = 1;
    for (const auto& l : lits)
        for (unsigned char c : l)
            if (!cmap[c]) cmap[c] = uint8_t(ncls++);
    std::vector<unsigned> rep(ncls, 256);  // (class 0 has no representative: it leads to the root from everywhere)
    for (unsigned b = 0; b < 256; ++b)
        if (cmap[b]) rep[cmap[b]] = b;
    // trie
    std::vector<std::vector<int32_t>> go(1, std::vector<int32_t>(ncls, -1));
    std::vector<uint64_t> out(1, 0);
    uint64_t always = 0;
    for (size_t p = 0; p < lits.size(); ++p) {
        if (lits[p].empty()) {
            always |= uint64_t(1) << p;
            continue;
        }
        int32_t s = 0;
        for (unsigned char c : lits[p]) {
            const uint32_t k = cmap[c];
            if (go[size_t(s)][k] < 0) {
                go[size_t(s)][k] = int32_t(go.size());
                go.emplace_back(ncls, -1);
                out.push_back(0);
            }
            s = go[size_t(s)][k];
        }
        out[size_t(s)] |= uint64_t(1) << p;
    }
    // failure links, breadth first; goto completed into a DFA
    const size_t nStates = go.size();
    std::vector<int32_t> fail(nStates, 0), order;
    for (uint32_t k = 0; k < ncls; ++k) {
        if (go[0][k] < 0) go[0][k] = 0;
        else order.push_back(go[0][k]);
    }
    for (size_t i = 0; i < order.size(); ++i) {
        const int32_t s = order[i];
        out[size_t(s)] |= out[size_t(fail[size_t(s)])];
        for (uint32_t k = 0; k < ncls; ++k) {
            const int32_t t = go[size_t(s)][k];
            if (t < 0) {
                go[size_t(s)][k] = go[size_t(fail[size_t(s)])][k];
            } else {
                fail[size_t(t)] = go[size_t(fail[size_t(s)])][k];
                order.push_back(t);
            }
        }
    }
    if (nStates > 0x7FFF) return {};
    std::vector<uint8_t> bytes(GL_HEADER_WORDS * 4 + 256, 0);
    std::memcpy(bytes.data() + GL_HEADER_WORDS * 4, cmap.data(), 256);
    auto append = [&](const void* p, size_t n) {
        const size_t at = (bytes.size() + 15) & ~size_t(15);
        bytes.resize(at + n);
        std::memcpy(bytes.data() + at, p, n);
        return uint32_t(at);
    };
    uint32_t hdr[GL_HEADER_WORDS] = {};
    hdr[GL_NSTATES] = uint32_t(nStates);
    hdr[GL_NCLASSES] = ncls;
    hdr[GL_OFF_MASKS] = append(out.data(), out.size() * 8);
    std::vector<uint16_t> table(nStates * ncls);
    for (size_t s = 0; s < nStates; ++s)
        for (uint32_t k = 0; k < ncls; ++k) {
            const int32_t t = go[s][k];
            table[s * ncls + k] = uint16_t(uint32_t(t) | (out[size_t(t)] ? 0x8000u : 0u));
        }
    hdr[GL_OFF_TABLE] = append(table.data(), table.size() * 2);
    hdr[GL_ALWAYS_LO] = uint32_t(always);
    hdr[GL_ALWAYS_HI] = uint32_t(always >> 32);
    std::memcpy(bytes.data(), hdr, sizeof hdr);
    bytes.resize((bytes.size() + 15) & ~size_t(15));
    std::vector<uint32_t> blob(bytes.size() / 4);
    std::memcpy(blob.data(), bytes.data(), bytes.size());
    return blob;
}
