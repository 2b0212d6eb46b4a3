#!/bin/bash
# effective clock of the wide GEMMs under the two schedules: GRBM_GUI_ACTIVE / duration, same box, same process order
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$PWD
AB=$PWD/easy_vitpose_amd/_lib/ab
cd /tmp && export TMPDIR=/tmp
for L in sched1 product sched1 product; do
  if [ $L = product ]; then LIB=$ROOT/easy_vitpose_amd/_lib/libvitpose_hip.so; else LIB=$AB/$L.so; fi
  OUT=$ROOT/gpurun_out/clk_$L
  rm -rf $OUT; mkdir -p $OUT
  VP_HIP_LIB=$LIB rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $OUT -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-host-path --steps 2 --warmup 1 > $OUT/log.txt 2>&1
  python - $OUT $L <<'PY'
import glob,sqlite3,sys,os
root,tag=sys.argv[1],sys.argv[2]
for db in glob.glob(os.path.join(root,'**','*.db'),recursive=True):
    cur=sqlite3.connect(db).cursor()
    rows=cur.execute('select kernel_name,counter_name,sum(value),count(distinct dispatch_id),avg(duration) from counters_collection group by kernel_name,counter_name').fetchall()
    per={}
    for k,c,v,n,dur in rows: per.setdefault(k,{'n':n,'dur':dur})[c]=v/n
    for k,d in sorted(per.items(), key=lambda kv:-kv[1]['dur']*kv[1]['n'])[:4]:
        name=k.replace('vp::','').replace('(anonymous namespace)::','')[:46]
        gui=d.get('GRBM_GUI_ACTIVE',0)/8
        print(f"{tag:8s} {name:46s} avg {d['dur']/1e3:7.1f} us  GRBM_GUI_ACTIVE/8 = {gui:9.0f} cycles -> {gui/d['dur']:.3f} GHz   MFMA busy/1024 = {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024:9.0f}  = {d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/max(gui,1):.3f} of the cycles")
PY
  rm -rf $OUT/*/
done
