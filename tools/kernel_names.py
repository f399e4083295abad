"""Map a rocprofv3 kernel name to the short name bench.py / Plan.kernel_of use."""
import re


def short(name: str) -> str:
    m = re.search(r'(\w+)<([^>]*)>\(', name) or re.search(r'(\w+)<([^>]*)>', name)
    if not m:
        m2 = re.search(r'::(\w+)\(', name) or re.search(r'(\w+)\(', name)
        return m2.group(1) if m2 else name
    fn, args = m.group(1), [a.strip() for a in m.group(2).split(',')]
    if fn == 'conv_igemm_glds_kernel':      # <BN, BM, WN, WM, BK, UT>
        k64 = ',K64' if len(args) > 4 and args[4] == '64' else ''
        return f'conv_igemm_glds_kernel<{args[0]},{args[1]}{k64}>'
    if fn == 'conv_igemm_kernel':
        return f'conv_igemm_kernel<{args[0]},{args[1]}>'
    if fn in ('conv3x3_tap_kernel', 'conv3x3_halo_kernel'):
        return f'{fn}<{args[0]}>' if fn == 'conv3x3_tap_kernel' else fn
    if fn in ('convpair2_kernel', 'convpair_kernel'):
        return f'{fn}<{args[0]},{args[1]}>'
    return fn
