#!/usr/bin/env python3
"""Symbolise a glog-style abort trace ("    @     0x7c7b... (unknown)") against the /proc/<pid>/maps of the SAME process
(bench.py writes it with TK_BENCH_DUMP_MAPS=path).  For every frame: the owning DSO, the file offset, and — when the DSO exists on this
machine (the GPU box runs the same image) — the nearest dynamic / static symbol below the address (nm -D --defined-only, nm for static).
usage: tools/symbolise_crash.py <trace.err> <maps.txt>"""
import bisect
import re
import subprocess
import sys


def load_maps(path):
    out = []
    for line in open(path):
        m = re.match(r"([0-9a-f]+)-([0-9a-f]+) (\S+) ([0-9a-f]+) \S+ \d+\s*(.*)", line)
        if m:
            out.append((int(m.group(1), 16), int(m.group(2), 16), m.group(3), int(m.group(4), 16), m.group(5).strip()))
    return out


_syms = {}


def symbols(dso):
    if dso in _syms:
        return _syms[dso]
    tab = []
    for args in (["nm", "-D", "--defined-only", "-C", dso], ["nm", "--defined-only", "-C", dso]):
        try:
            txt = subprocess.run(args, capture_output=True, text=True, timeout=120).stdout
        except Exception:
            continue
        for ln in txt.splitlines():
            p = ln.split(None, 2)
            if len(p) == 3 and p[1] in "TtWwiV":
                try:
                    tab.append((int(p[0], 16), p[2]))
                except ValueError:
                    pass
    tab = sorted(set(tab))
    _syms[dso] = tab
    return tab


def vaddr_of(dso, file_off):
    """file offset -> link-time virtual address via the PT_LOAD headers (readelf -lW)"""
    try:
        txt = subprocess.run(["readelf", "-lW", dso], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return file_off
    for ln in txt.splitlines():
        p = ln.split()
        if p and p[0] == "LOAD":
            off, va, fsz = int(p[1], 16), int(p[2], 16), int(p[4], 16)
            if off <= file_off < off + fsz:
                return va + (file_off - off)
    return file_off


def main():
    trace, maps = sys.argv[1], load_maps(sys.argv[2])
    addrs = []
    for ln in open(trace):
        m = re.search(r"(PC: )?@\s+0x([0-9a-f]+)\s+(.*)", ln)
        if m:
            addrs.append((int(m.group(2), 16), m.group(3).strip(), bool(m.group(1))))
        m = re.search(r"SIGSEGV \(@0x([0-9a-f]+)\)", ln)
        if m:
            fa = int(m.group(1), 16)
            print("fault address 0x%x:" % fa)
            for lo, hi, perm, off, name in maps:
                if lo - 0x200000 <= fa < hi + 0x200000:
                    print("   %s 0x%x-0x%x %s %s%s" % ("->" if lo <= fa < hi else "  ", lo, hi, perm, name or "[anon]", "   (ends exactly at the fault address)" if hi == fa else ""))
    for a, given, is_pc in addrs:
        hit = next(((lo, hi, perm, off, name) for lo, hi, perm, off, name in maps if lo <= a < hi), None)
        if not hit:
            print("0x%x  <unmapped>  %s" % (a, given))
            continue
        lo, hi, perm, off, name = hit
        fo = a - lo + off
        sym = ""
        if name.startswith("/"):
            va = vaddr_of(name, fo)
            tab = symbols(name)
            i = bisect.bisect_right(tab, (va, "\xff")) - 1
            if i >= 0:
                sym = "%s+0x%x" % (tab[i][1], va - tab[i][0])
        print("%s0x%x  %s +0x%x  %s  %s" % ("PC " if is_pc else "   ", a, name.split("/")[-1] or "[anon]", fo, sym, "" if given == "(unknown)" else "[" + given + "]"))


if __name__ == "__main__":
    main()
