#!/bin/bash
# Build recipe for the REAL reference (PaStiX 5.2.2.16 CPU sopalin) -> oracle/_ref/
#
# TEST INFRASTRUCTURE ONLY.  Compiles the reference sources where they lie under
# /root/reference (never copied into this repo) with plain gcc + the image's MKL,
# following the source lists of the reference's own per-module CMakeLists.txt
# (src/*/src/CMakeLists.txt) and the 4 factorization variants of SRC_FAC
# (src/CMakeLists.txt:37-116, sopalin_define.h:453-465).
# Outputs only into oracle/_ref/ (git-ignored; travels to the GPU box via gpurun).
#
# usage: oracle/build_ref.sh [PREC] [BLAS]   PREC = d (default) | z ;  BLAS = mkl (default) | openblas
#   openblas = the OpenBLAS that ships inside the image's scipy wheel (LP64, symbols prefixed
#   scipy_): same reference sources, BLAS symbols renamed on the compiler command line.  It is the
#   build used for the CPU baseline on AMD hosts, where MKL's dispatcher picks a slow generic path.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${PASTIX_REFERENCE:-/root/reference}
PREC=${1:-d}
BLAS=${2:-mkl}
OUT="$HERE/_ref"
SUF=""
[ "$BLAS" = openblas ] && SUF="_ob"
OBJ="$OUT/obj_$PREC$SUF"
[ -d "$REF/src" ] || { echo "reference not present at $REF: keeping prebuilt oracle/_ref"; exit 0; }
MKL=${MKL_LIBDIR:-/opt/conda/lib}
[ -e "$MKL/libmkl_rt.so" ] || { echo "no MKL at $MKL: reference unbuildable here"; exit 0; }
OB=$(ls /usr/local/lib/python3*/dist-packages/scipy.libs/libscipy_openblas*.so 2>/dev/null | head -1)
if [ "$BLAS" = openblas ]; then
  [ -n "$OB" ] || { echo "no scipy OpenBLAS in this image"; exit 0; }
  for f in scal axpy copy gemm trsv trsm syrk syr ger geru gerc her herk gemv swap dot nrm2; do
    RENAMES="$RENAMES -Dd${f}_=scipy_d${f}_ -Dz${f}_=scipy_z${f}_"
  done
fi
mkdir -p "$OBJ"
S="$REF/src"
INC="-I$S/common/src -I$S/symbol/src -I$S/order/src -I$S/fax/src -I$S/kass/src -I$S/blend/src -I$S/sopalin/src -I$S/perf/src -I$S/sparse-matrix/src -I$S/matrix_drivers/src"
DEFS="-DFORCE_NOMPI -DPREC_DOUBLE -DINTSIZE32 -DVERSION=\"ref\" -DX_ARCHi686_pc_linux -DDOF_CONSTANT -DFORCE_NO_CUDA"
[ "$PREC" = z ] && DEFS="$DEFS -DTYPE_COMPLEX"
DEFS="$DEFS $RENAMES"
CC="gcc -O2 -fPIC -w -std=gnu99 $INC $DEFS"

COMMON="common_integer common_error common_memory trace common"
SYMBOL="dof dof_io symbol symbol_base symbol_check symbol_cost symbol_draw symbol_io symbol_keep symbol_levf symbol_nonzeros symbol_tree"
ORDER="order order_base order_check order_io"
BLEND="assemblyGener blend blend_symbol_cost blendctrl bulles cost costfunc distribPart elimin eliminfunc extendVector extrastruct fanboth2 param_blend partbuild queue simu smart_cblk_split solverMatrixGen solverRealloc solver_check solver_io splitfunc splitpart splitpartlocal symbolrand task write_ps blend_distributeOnGPU"
FAX="symbol_compact symbol_costi symbol_fax_graph symbol_fax symbol_faxi_graph symbol_faxi"
KASS="kass compact_graph amalgamate ifax sparRow SF_Direct SF_level find_supernodes KSupernodes sort_row"
SPM="pastix_sparse_matrix"
SOPALIN="bordi sopalin_thread compute_context_nbr coefinit csc_intern_build csc_intern_io csc_intern_solve csc_intern_updown csc_utils cscd_utils cscd_utils_fortran debug_dump ooc pastix pastix_fortran sopalin_init sopalin_option sparse_gemm_cpu tools"
FAC="sopalin3d starpu_submit_tasks csc_intern_compute raff_functions starpu_updo"

pids=()
cc() { # dir name extra-defs suffix
  local o="$OBJ/$2$4.o"
  [ "$o" -nt "$S/$1/src/$2.c" ] && return 0
  $CC $3 -c "$S/$1/src/$2.c" -o "$o" &
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait "${pids[@]}"; pids=(); fi
}
for f in $COMMON; do cc common $f -DCHOL_SOPALIN ""; done
for f in $SYMBOL; do cc symbol $f -DCHOL_SOPALIN ""; done
for f in $ORDER;  do cc order  $f -DCHOL_SOPALIN ""; done
for f in $BLEND;  do cc blend  $f -DCHOL_SOPALIN ""; done
for f in $FAX;    do cc fax    $f -DCHOL_SOPALIN ""; done
for f in $KASS;   do cc kass   $f -DCHOL_SOPALIN ""; done
for f in $SPM;    do cc sparse-matrix $f -DCHOL_SOPALIN ""; done
for f in $SOPALIN; do cc sopalin $f -DCHOL_SOPALIN ""; done
for f in $FAC; do
  cc sopalin $f "-DCHOL_SOPALIN" "_po"
  cc sopalin $f "-DCHOL_SOPALIN -DSOPALIN_LU" "_ge"
  cc sopalin $f "" "_sy"
  cc sopalin $f "-DHERMITIAN" "_he"
done
wait
# the harness driver (ours) includes the reference's internal headers to read SolverMatrix
$CC -DCHOL_SOPALIN -c "$HERE/ref_harness.c" -o "$OBJ/ref_harness.o"
if [ "$BLAS" = openblas ]; then
  gcc -o "$OUT/ref_harness_$PREC$SUF" "$OBJ"/*.o "$OB" -Wl,-rpath,"$(dirname "$OB")" -lpthread -lm -lrt
else
  gcc -o "$OUT/ref_harness_$PREC" "$OBJ"/*.o -L"$MKL" -Wl,-rpath,"$MKL" -lmkl_rt -lpthread -lm -lrt
fi
echo "built $OUT/ref_harness_$PREC$SUF"

# ---- the reference as the CALLER of the MI355X engine (oracle/ref_amd_sopalin3d.c + integration/sopalin_amd_stub.h):
# same objects, except that the four sopalin3d variants are rebuilt from the wrapper TU, which includes the reference's
# sopalin3d.c from where it lies and redirects its sopalin_launch_thread() calls to the stub.
LIBAMD="$HERE/../pastix_amd/lib"
# (OpenBLAS build only: the MKL of this image sits beside an older libstdc++ that libpastix_amd.so / the HIP runtime
# cannot run with)
if [ -e "$LIBAMD/libpastix_amd.so" ] && [ "$BLAS" = openblas ]; then
  AOBJ="$OBJ/amd"
  mkdir -p "$AOBJ"
  INCAMD="-I$HERE/../include"
  $CC $INCAMD -DCHOL_SOPALIN -DAMD_HOOK=pastix_amd_launch_po -c "$HERE/ref_amd_sopalin3d.c" -o "$AOBJ/sopalin3d_po.o" &
  $CC $INCAMD -DCHOL_SOPALIN -DSOPALIN_LU -DAMD_HOOK=pastix_amd_launch_ge -c "$HERE/ref_amd_sopalin3d.c" -o "$AOBJ/sopalin3d_ge.o" &
  $CC $INCAMD -DAMD_HOOK=pastix_amd_launch_sy -c "$HERE/ref_amd_sopalin3d.c" -o "$AOBJ/sopalin3d_sy.o" &
  $CC $INCAMD -DHERMITIAN -DAMD_HOOK=pastix_amd_launch_he -c "$HERE/ref_amd_sopalin3d.c" -o "$AOBJ/sopalin3d_he.o" &
  wait
  REST=$(ls "$OBJ"/*.o | grep -v -E "/sopalin3d_(po|ge|sy|he)\.o$")
  RP="-Wl,-rpath,\$ORIGIN/../../pastix_amd/lib -Wl,-rpath,/opt/rocm/lib"
  gcc -o "$OUT/ref_harness_$PREC${SUF}_amd" $REST "$AOBJ"/*.o "$OB" -Wl,-rpath,"$(dirname "$OB")" -L"$LIBAMD" -lpastix_amd $RP -lpthread -lm -lrt
  echo "built $OUT/ref_harness_$PREC${SUF}_amd (reference pastix() -> blend -> MI355X engine -> reference updo)"
fi
