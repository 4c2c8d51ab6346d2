"""Kernel resource table of a code object: python profiles/kres.py <readelf --notes output> -> registers, LDS, scratch per kernel."""
import re, subprocess, sys
t = open(sys.argv[1]).read()
for k in re.split(r'\n\s+- \.agpr_count', t)[1:]:
    name = re.search(r'\.name:\s+(\S+)', k).group(1)
    g = lambda f: re.search(r'\.%s:\s+(\d+)' % f, k).group(1)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    m = re.search(r'(\w+)<([^>]*)>', dn)
    print((m.group(1) + '<' + m.group(2) + '>') if m else dn[:80], 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'), 'spill', g('vgpr_spill_count'))
