cd $GRAFT_REPO_ROOT
python - <<'PY'
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
from easy_vitpose_amd import _capi as capi
lib = capi.load_library()
M = 49152
for name, epi, N, K, va, vb, fl in (('qkv', 0, 2304, 768, 16, 20, 16), ('fc1', 1, 3072, 768, 16, 20, 18), ('fc2', 6, 768, 3072, 17, 21, 12)):
    nm, md = C.c_uint64(), C.c_double()
    rc = lib.vp_dbg_gemm_compare(0, 0, epi, va, 8, fl, vb, 8, fl, M, N, K, 2, C.byref(nm), C.byref(md))
    print(name, 'compare split vs plain: rc', rc, 'mismatches', nm.value, capi.last_error() if rc else '')
    for rnd in range(3):
        row=[]
        for v in (va, vb):
            ms = C.c_float()
            rc = lib.vp_dbg_gemm_bench2(0, 0, epi, v, 8, fl, M, N, K, 10, C.byref(ms))
            row.append(f'v{v}: {ms.value*1e3:.1f}')
        print(name, ' '.join(row), flush=True)
PY
