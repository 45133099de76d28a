// pt_isa_hash.cpp — the hash of the device assembly the library's kernels were built from (Makefile: the first 16 hex digits
// of sha256 over pt_kernels.s, the -save-temps listing of the very compile that produced the kernels' object).
// profiles/*_traffic.json carry the hash of the library they were measured on; bench.py compares (profile_matches_binary).
// Diagnostic one-shot builds (make phase / walk / variant) have no listing of their own and report "unknown".
#ifndef PT_KERNEL_ISA_HASH
#define PT_KERNEL_ISA_HASH "unknown"
#endif
extern "C" const char *pt_kernel_isa_hash(void) { return PT_KERNEL_ISA_HASH; }
