import re,sys,subprocess,os
src=sys.argv[1]
out='/tmp/kmeta_'+os.path.basename(src)+'.s'
subprocess.check_call(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-std=c++17','-fPIC','-ffp-contract=fast','-Wno-unused-result','-DNDEBUG','-w','-S','--cuda-device-only',src,'-o',out])
s=open(out).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    name=m.group(1); body=m.group(2)
    g=lambda k: (re.search(r'\.'+k+r':\s+(\d+)',body) or [None,None])[1]
    dem=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    dem=re.sub(r'\(anonymous namespace\)::','',dem)[:110]
    print(f"{dem:110s} vgpr {g('vgpr_count')} spill {g('vgpr_spill_count')} sgpr {g('sgpr_count')} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")
