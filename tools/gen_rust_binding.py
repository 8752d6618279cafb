#!/usr/bin/env python3
"""Generates the Rust `-sys` binding (extern "C" block + #[repr(C)] structs) from include/rdoom.h, so that the text in
INTEGRATION.md cannot drift from the header: tests/test_c_abi.py regenerates it and compares.

    python tools/gen_rust_binding.py            # prints the binding
    python tools/gen_rust_binding.py --update   # rewrites the block between the markers in INTEGRATION.md

The header is written in a small, regular subset of C (typedef struct { ... } name; opaque typedefs; #define constants;
anonymous enums; prototypes returning rdoom_status / void / const char *), which is all this translator understands --
it fails loudly on anything else.  No Rust toolchain exists in the build image; the output has never been compiled here."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'rdoom.h')
DOC = os.path.join(ROOT, 'INTEGRATION.md')
BEGIN, END = '<!-- BEGIN GENERATED rdoom-sys -->', '<!-- END GENERATED rdoom-sys -->'

SCALARS = {'float': 'f32', 'double': 'f64', 'uint8_t': 'u8', 'uint16_t': 'u16', 'uint32_t': 'u32', 'uint64_t': 'u64',
           'int8_t': 'i8', 'int16_t': 'i16', 'int32_t': 'i32', 'int64_t': 'i64', 'char': 'c_char', 'void': 'c_void',
           'rdoom_status': 'rdoom_status'}


def strip_comments(text):
    return re.sub(r'/\*.*?\*/', '', text, flags=re.S)


def rust_type(ctype, names):
    """`const T *`, `T **`, `const T *const *`, `T` -> Rust; names = struct / opaque types the header defines"""
    t = ' '.join(ctype.replace('*', ' * ').split())
    toks = t.split(' ')
    const = toks[0] == 'const'
    base = toks[1] if const else toks[0]
    if base not in SCALARS and base not in names:
        raise SystemExit('gen_rust_binding: unknown C type %r' % ctype)
    r = SCALARS.get(base, base)
    # every `*` makes a pointer to what stands left of it; a `const` right after a `*` qualifies THAT pointer, i.e. what the
    # next `*` points at
    pointee_const = const
    for i, tok in enumerate(toks[2 if const else 1:], start=2 if const else 1):
        if tok == '*':
            r = ('*const ' if pointee_const else '*mut ') + r
            pointee_const = i + 1 < len(toks) and toks[i + 1] == 'const'
        elif tok != 'const':
            raise SystemExit('gen_rust_binding: unknown C type %r' % ctype)
    return r


def parse(text):
    text = strip_comments(text)
    consts = re.findall(r'^#define\s+(RDOOM_[A-Z0-9_]+)\s+\(?(-?(?:0x[0-9A-Fa-f]+|\d+))u?\)?\s*$', text, flags=re.M)
    opaque = re.findall(r'^typedef struct (\w+) (\w+);\s*$', text, flags=re.M)
    structs = re.findall(r'typedef struct (\w+) \{(.*?)\} (\w+);', text, flags=re.S)
    enums = re.findall(r'enum \{(.*?)\};', text, flags=re.S)
    protos = re.findall(r'^((?:const\s+)?\w+\s*\*?)\s*(rdoom_\w+)\s*\(([^;{]*?)\);', text, flags=re.M | re.S)
    return consts, [o[1] for o in opaque], structs, enums, protos


def struct_fields(body, names):
    out = []
    for decl in [d.strip() for d in body.split(';') if d.strip()]:
        decl = ' '.join(decl.split())
        m = re.match(r'^(.*?)\(\*(\w+)\)\((.*)\)$', decl)   # function pointer member
        if m:
            ret, fname, args = m.group(1).strip(), m.group(2), m.group(3)
            assert ret == 'void', decl
            out.append((fname, 'Option<unsafe extern "C" fn(%s)>' % ', '.join('%s: %s' % a for a in params(args, names))))
            continue
        m = re.match(r'^((?:const\s+)?\w+(?:\s*\*+)?)\s*(.*)$', decl)
        ctype, rest = m.group(1), m.group(2)
        for item in [i.strip() for i in rest.split(',')]:
            stars = len(item) - len(item.lstrip('*'))
            item = item.lstrip('* ')
            am = re.match(r'^(\w+)\[(\d+)\]$', item)
            t = rust_type(ctype + '*' * stars, names)
            out.append((am.group(1), '[%s; %s]' % (t, am.group(2))) if am else (item, t))
    return out


def params(args, names):
    args = ' '.join(args.split())
    if args in ('', 'void'):
        return []
    out = []
    for a in [x.strip() for x in args.split(',')]:
        am = re.match(r'^(.*?)(\w+)\[(\d*)\]$', a)   # `float out[3]` decays to a pointer
        if am:
            out.append((am.group(2), rust_type(am.group(1) + '*', names)))
            continue
        m = re.match(r'^(.*?)(\w+)$', a)
        out.append((m.group(2), rust_type(m.group(1), names)))
    return out


def generate():
    consts, opaque, structs, enums, protos = parse(open(HEADER).read())
    names = set(opaque) | {s[2] for s in structs}
    lines = ['// rdoom-sys/src/lib.rs -- GENERATED from include/rdoom.h by tools/gen_rust_binding.py; do not edit',
             '#![allow(non_camel_case_types)]', 'use std::os::raw::{c_char, c_void};', '', 'pub type rdoom_status = i32;']
    for name, value in consts:
        ty = 'rdoom_status' if int(value, 0) <= 0 and 'KIND' not in name else 'u32'
        lines.append('pub const %s: %s = %s;' % (name, ty, value))
    for body in enums:
        for item in [i.strip() for i in body.split(',') if i.strip()]:
            k, v = [x.strip() for x in item.split('=')]
            lines.append('pub const %s: i32 = %s;' % (k, v))
    lines.append('')
    for name in opaque:
        if name not in {s[2] for s in structs}:
            lines.append('#[repr(C)] pub struct %s { _private: [u8; 0] }' % name)
    for _tag, body, name in structs:
        fields = struct_fields(body, names)
        copy = not any('fn(' in t for _, t in fields)
        lines.append('')
        lines.append('#[repr(C)]%s' % (' #[derive(Copy, Clone)]' if copy else ''))
        lines.append('pub struct %s {' % name)
        lines.extend('    pub %s: %s,' % f for f in fields)
        lines.append('}')
    lines += ['', '#[link(name = "rdoom_hip")]', 'extern "C" {']
    for ret, name, args in protos:
        ret = ' '.join(ret.split())
        r = '' if ret == 'void' else ' -> ' + rust_type(ret, names)
        lines.append('    pub fn %s(%s)%s;' % (name, ', '.join('%s: %s' % p for p in params(args, names)), r))
    lines.append('}')
    return '\n'.join(lines) + '\n'


def main():
    text = generate()
    if '--update' in sys.argv:
        doc = open(DOC).read()
        a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
        open(DOC, 'w').write(doc[:a] + '\n```rust\n' + text + '```\n' + doc[b:])
    else:
        sys.stdout.write(text)


if __name__ == '__main__':
    main()
