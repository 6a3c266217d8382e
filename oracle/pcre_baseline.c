/* oracle/pcre_baseline.c -- TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/README.md): never linked by the product.
 *
 * BASELINE.md section 2 names PCRE1 8.45 as the stand-in engine for boost::regex_match in the CPU baseline (boost is
 * neither in this image nor on the GPU box: profiles/round2_gpu_box_probe.txt).  This file times that engine on the same
 * batch layout the oracle uses: pattern wrapped as (?:re)\z, PCRE_DOTALL | PCRE_MULTILINE | anchored = the full-match call
 * core/common/StringTools.cpp:183-211 makes.  libpcre is opened at run time (no headers in the image's system include
 * path are required); the prototypes below are PCRE1's published C API (pcre.h 8.x).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct real_pcre pcre;
typedef struct pcre_extra pcre_extra;
typedef pcre* (*pcre_compile_fn)(const char*, int, const char**, int*, const unsigned char*);
typedef pcre_extra* (*pcre_study_fn)(const pcre*, int, const char**);
typedef int (*pcre_exec_fn)(const pcre*, const pcre_extra*, const char*, int, int, int, int*, int);
typedef const char* (*pcre_version_fn)(void);

#define PCRE_MULTILINE 0x00000002
#define PCRE_DOTALL 0x00000004
#define PCRE_ANCHORED 0x00000010
#define PCRE_STUDY_JIT_COMPILE 0x0001

static void* gLib;
static pcre_compile_fn gCompile;
static pcre_study_fn gStudy;
static pcre_exec_fn gExec;
static pcre_version_fn gVersion;

static int openPcre(void) {
    if (gLib) return 1;
    const char* names[] = {"/opt/conda/lib/libpcre.so.1", "libpcre.so.3", "libpcre.so.1", "libpcre.so"};
    for (unsigned i = 0; i < sizeof names / sizeof names[0] && !gLib; ++i) gLib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!gLib) return 0;
    gCompile = (pcre_compile_fn)dlsym(gLib, "pcre_compile");
    gStudy = (pcre_study_fn)dlsym(gLib, "pcre_study");
    gExec = (pcre_exec_fn)dlsym(gLib, "pcre_exec");
    gVersion = (pcre_version_fn)dlsym(gLib, "pcre_version");
    return gCompile && gStudy && gExec;
}

/* "8.45 2021-06-15" or "" when no PCRE1 could be opened */
const char* orx_pcre_version(void) { return openPcre() && gVersion ? gVersion() : ""; }

/* Full-matches n lines (data + off[i], len[i]) and writes caps[n][2*ngroups] / status[n] in the oracle's layout.
 * jit != 0: pcre_study(PCRE_STUDY_JIT_COMPILE).  Returns the number of matching lines, -1: no libpcre, -2: compile error. */
long orx_pcre_fullmatch_batch(const char* pattern, size_t patternLen, const uint8_t* data, const uint32_t* off,
                              const uint32_t* len, size_t n, int ngroups, int jit, int32_t* caps, uint8_t* status) {
    if (!openPcre()) return -1;
    char* wrapped = (char*)malloc(patternLen + 8);
    memcpy(wrapped, "(?:", 3);
    memcpy(wrapped + 3, pattern, patternLen);
    memcpy(wrapped + 3 + patternLen, ")\\z", 4);
    const char* err = NULL;
    int erroff = 0;
    pcre* code = gCompile(wrapped, PCRE_DOTALL | PCRE_MULTILINE, &err, &erroff, NULL);
    free(wrapped);
    if (!code) return -2;
    pcre_extra* extra = gStudy(code, jit ? PCRE_STUDY_JIT_COMPILE : 0, &err);
    const int ovn = 3 * (ngroups + 1);
    int* ov = (int*)malloc(sizeof(int) * (size_t)ovn);
    long matched = 0;
    for (size_t i = 0; i < n; ++i) {
        const int rc = gExec(code, extra, (const char*)data + off[i], (int)len[i], 0, PCRE_ANCHORED, ov, ovn);
        int32_t* row = caps + i * 2 * (size_t)ngroups;
        if (rc < 0) {
            status[i] = 0;
            for (int g = 0; g < 2 * ngroups; ++g) row[g] = -1;
            continue;
        }
        status[i] = 1;
        ++matched;
        for (int g = 1; g <= ngroups; ++g) {
            const int set = rc == 0 || g < rc;
            row[2 * g - 2] = set ? ov[2 * g] : -1;
            row[2 * g - 1] = set ? ov[2 * g + 1] : -1;
        }
    }
    free(ov);
    /* (pcre_free / pcre_free_study are data symbols holding function pointers; the few KB are left to process exit) */
    return matched;
}
