"""Code bytes of every kernel in libmpg_hip.so (the text each CU has to fetch cold once per launch):
    python3 tools/code_size.py [substring ...]"""
import os, subprocess, sys, tempfile, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
so = os.path.join(ROOT, 'mpg_amd', 'libmpg_hip.so')
with tempfile.TemporaryDirectory() as d:
    # the fat binary section holds one bundle per translation unit
    subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + d + '/fat.bin', so], check=True)
    blob = open(d + '/fat.bin', 'rb').read()
    out = []
    pos = 0
    k = 0
    while True:
        i = blob.find(b'\x7fELF', pos)
        if i < 0: break
        # e_shoff + e_shnum * e_shentsize bounds the ELF
        import struct
        e_shoff = struct.unpack_from('<Q', blob, i + 0x28)[0]
        e_shentsize, e_shnum = struct.unpack_from('<HH', blob, i + 0x3A)
        end = i + e_shoff + e_shentsize * e_shnum
        path = '%s/co%d.elf' % (d, k); k += 1
        open(path, 'wb').write(blob[i:end])
        r = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-s', '--wide', path], capture_output=True, text=True).stdout
        for line in r.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == 'FUNC':
                out.append((int(f[2]), f[7]))
        pos = end
    dem = subprocess.run(['c++filt'], input='\n'.join(n for _, n in out), capture_output=True, text=True).stdout.splitlines()
    rows = sorted(set((s, n) for (s, _), n in zip(out, dem)), reverse=True)
    for s, n in rows:
        n = n.replace('(anonymous namespace)::', '')
        n = re.sub(r'\((mlp::|rollout::|int,|float|const)[^)]*\).*$', '', n)
        if len(sys.argv) > 1 and not any(a in n for a in sys.argv[1:]): continue
        print('%8d B  %5.1f KB  %s' % (s, s / 1024.0, n))
