"""Kernel metadata of the gfx950 code objects inside libllv.so (no ROCm tool needed: ELF + clang offload bundle + the
NT_AMDGPU_METADATA msgpack note are parsed here).

  python tools/code_objects.py [path/to/libllv.so]      # table: private segment (scratch), LDS, VGPR, SGPR, spills per kernel

tests/test_abi.py uses kernels_metadata() to assert that no kernel of the library needs scratch memory or spills."""
import struct
import sys

import msgpack

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def _elf_sections(data):
    assert data[:4] == b'\x7fELF' and data[4] == 2, 'ELF64 expected'
    shoff, = struct.unpack_from('<Q', data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', data, 0x3A)
    secs = []
    for i in range(shnum):
        o = shoff + i * shentsize
        name, typ, flags, addr, off, size = struct.unpack_from('<IIQQQQ', data, o)
        secs.append(dict(name_off=name, type=typ, off=off, size=size))
    stro = secs[shstrndx]['off']
    for s in secs:
        e = data.index(b'\0', stro + s['name_off'])
        s['name'] = data[stro + s['name_off']:e].decode()
    return secs


def _bundles(fat):
    """(target id, code object bytes) of every entry of every bundle in a .hip_fatbin section"""
    pos = 0
    while True:
        pos = fat.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from('<Q', fat, pos + len(MAGIC))
        o = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from('<QQQ', fat, o)
            tid = fat[o + 24:o + 24 + idlen].decode()
            o += 24 + idlen
            yield tid, fat[pos + off:pos + off + size]
        pos += len(MAGIC)


def _notes(elf):
    for s in _elf_sections(elf):
        if s['type'] != 7:          # SHT_NOTE
            continue
        d, o, end = elf, s['off'], s['off'] + s['size']
        while o + 12 <= end:
            namesz, descsz, typ = struct.unpack_from('<III', d, o)
            o += 12
            name = d[o:o + namesz]
            o += (namesz + 3) & ~3
            desc = d[o:o + descsz]
            o += (descsz + 3) & ~3
            yield name.rstrip(b'\0'), typ, desc


def kernels_metadata(lib_path):
    """[{name, private_segment_fixed_size, group_segment_fixed_size, vgpr_count, sgpr_count, vgpr_spill_count, ...}] for
    every kernel of every gfx950 code object bundled into the shared library."""
    data = open(lib_path, 'rb').read()
    fat = None
    for s in _elf_sections(data):
        if s['name'] == '.hip_fatbin':
            fat = data[s['off']:s['off'] + s['size']]
    assert fat is not None, 'no .hip_fatbin section'
    out = []
    for tid, co in _bundles(fat):
        if 'gfx950' not in tid or not co:
            continue
        for name, typ, desc in _notes(co):
            if name == b'AMDGPU' and typ == 32:      # NT_AMDGPU_METADATA
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get('amdhsa.kernels', []):
                    out.append({key.lstrip('.'): v for key, v in k.items() if not isinstance(v, (list, dict))})
    return out


if __name__ == '__main__':
    import os
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              'lowlightenvironmentvideoobjectdetection_amd', 'libllv.so')
    ks = kernels_metadata(path)
    print(f'{len(ks)} kernels in {path}')
    for k in sorted(ks, key=lambda k: k['name']):
        print(f"scratch={k['private_segment_fixed_size']:5d} lds={k['group_segment_fixed_size']:6d} vgpr={k['vgpr_count']:3d} "
              f"sgpr={k['sgpr_count']:3d} spill={k['vgpr_spill_count']:3d}  {k['name'][:110]}")
