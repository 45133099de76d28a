// pt_kernels_flat.hip - the translation unit of k_pass_cand's instances WITHOUT walks (scenes whose meshes are all candidate
// records: the bench scene): the same source as pt_kernels.hip, of which PT_TU_FLAT leaves the kernel template, what it uses
// and launch_pass_cand_flat.  A unit of its own because the -mllvm options that steer instruction scheduling are per compile,
// and this kernel wants other ones than the kernels that walk (Makefile: MLLVM_FLAT).
#define PT_TU_FLAT 1
#include "pt_kernels.hip"
