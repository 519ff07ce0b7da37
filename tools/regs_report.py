"""Registers, spills and scratch per kernel from a hipcc log made with -Rpass-analysis=kernel-resource-usage."""
import re, sys
t = open(sys.argv[1]).read()
for b in re.split(r'remark: Function Name: ', t)[1:]:
    name = b.split(' ')[0]
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1))
    m = re.search(r'hdb_mfma_kernelI(\S+?)EEv', name)
    print((m.group(1) if m else name)[:60].ljust(60), 'VGPR', g('VGPRs'), 'AGPR', g('AGPRs'), 'spill', g('VGPRs Spill'), 'scratch', g(r'ScratchSize \[bytes/lane\]'), 'occ', g(r'Occupancy \[waves/SIMD\]'))
