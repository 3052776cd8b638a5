// What the library knows about its own build: the compiler, and whether the disassembly check of the hidden id-stream registers
// (check_hidden_regs.py, run by the Makefile on the freshly built enrich.o) passed.  obj/buildinfo.h is generated at build time.
#include <cstdio>
#include <cstring>

#include "common.h"
#include "draws.h"
#include "obj/buildinfo.h"

bool safe_hidden_regs_checked() { return SAFE_HIDDEN_REGS_CHECKED != 0; }

int safe_build_info(char *out, size_t out_len) {
    SAFE_REQUIRE(out && out_len, "safe_build_info: NULL argument");
#ifdef SAFE_HIP_DIAG
    const char *diag = "; DIAGNOSTIC build (variants that skip work are compiled in)";
#else
    const char *diag = "";
#endif
    snprintf(out, out_len, "%s; %s%s; seeded draw chain: %s", SAFE_BUILD_COMPILER, SAFE_BUILD_NOTE, diag, draws_path_name());
    return SAFE_OK;
}
