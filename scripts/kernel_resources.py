"""Register / LDS use of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python scripts/kernel_resources.py iterative_inference_segm_amd/csrc/conv_c8_bf16.hip [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-I' + root + '/include',
       '-I' + root + '/iterative_inference_segm_amd/csrc', '-c', src, '-o', '/dev/null',
       '-Rpass-analysis=kernel-resource-usage']
r = subprocess.run(cmd, capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    if 'error' in line:
        print(line)
    m = re.search(r'remark: +Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r'\(anonymous namespace\)::', '', cur)
        cur = re.sub(r'\(.*', '', cur)
        rows[cur] = {}
        continue
    m = re.search(r'remark: +([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print('%-70s VGPR %3d AGPR %3d spill %3d sgpr-spill %3d scratch %4d LDS %6d occ %s'
              % (k[-70:], v.get('VGPRs', -1), v.get('AGPRs', -1), v.get('VGPRs Spill', -1), v.get('SGPRs Spill', -1),
                 v.get('ScratchSize', -1), v.get('LDS Size', -1), v.get('Occupancy', '?')))
