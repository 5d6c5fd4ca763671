#!/usr/bin/env python3
"""Type-check the cgo call sites of go/gpuverifier against include/sbv.h without a Go toolchain (this image has none):
every `C.sbv_*(...)` call must name a function the header declares, pass as many arguments as it has parameters, pass a
pointer expression of the matching element type where the parameter is a pointer (u8(x) / (*C.uint8_t)(...) for uint8_t*,
(*C.uint32_t)(...) or &v with `var v C.uint32_t` for uint32_t*, (*C.uint64_t)(...) for uint64_t*, nil where the ABI allows
NULL) and a C.<type>(...) conversion or a literal where it is a scalar.  Exit status 0 = clean; problems go to stdout.
Used by tests/test_go_adapter.py."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_header(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|void\s*\*|const\s+char\s*\*|void|size_t)\s+(sbv_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                is_ptr = "*" in a or "[" in a
                base = re.sub(r"\bconst\b", "", a)
                base = re.sub(r"\[[^\]]*\]", "", base).replace("*", " ")
                toks = base.split()
                ctype = " ".join(toks[:-1]) if len(toks) > 1 else toks[0]      # drop the parameter name
                params.append((ctype, is_ptr))
        protos[name] = params
    return protos


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def calls(src):
    for m in re.finditer(r"C\.(sbv_\w+)\(", src):
        i, depth = m.end(), 1
        while depth and i < len(src):
            depth += src[i] == "("
            depth -= src[i] == ")"
            i += 1
        yield m.group(1), split_args(src[m.end():i - 1]), src.count("\n", 0, m.start()) + 1


def check(go_dir, header):
    protos = parse_header(header)
    problems, seen = [], 0
    for fn in sorted(os.listdir(go_dir)):
        if not fn.endswith(".go"):
            continue
        src = open(os.path.join(go_dir, fn)).read()
        cvars = dict(re.findall(r"var\s+(\w+)\s+C\.(\w+)", src))
        for name, args, line in calls(src):
            seen += 1
            where = f"{fn}:{line}: C.{name}"
            if name not in protos:
                problems.append(f"{where}: not declared in include/sbv.h")
                continue
            params = protos[name]
            if len(args) != len(params):
                problems.append(f"{where}: {len(args)} arguments, the header declares {len(params)}")
                continue
            for k, (arg, (ctype, is_ptr)) in enumerate(zip(args, params)):
                if is_ptr:
                    if arg == "nil":
                        continue
                    if ctype == "uint8_t":
                        ok = arg.startswith("u8(") or arg.startswith("(*C.uint8_t)(")
                    elif ctype in ("uint32_t", "uint64_t"):
                        ok = arg.startswith(f"(*C.{ctype})(") or (arg.startswith("&") and cvars.get(arg[1:]) == ctype)
                    elif ctype == "void":
                        ok = arg.startswith("unsafe.Pointer(")
                    else:
                        ok = arg.startswith("(*C.") or arg.startswith("&")
                    if not ok:
                        problems.append(f"{where}: argument {k + 1} `{arg}` is not a {ctype}* expression")
                else:
                    want = ctype.replace("unsigned ", "u")
                    ok = arg.startswith(f"C.{want}(") or re.fullmatch(r"-?\d+", arg) is not None
                    if not ok:
                        problems.append(f"{where}: argument {k + 1} `{arg}` should be C.{want}(...) or a literal")
    return seen, problems, protos


def main():
    seen, problems, protos = check(os.path.join(ROOT, "go", "gpuverifier"), os.path.join(ROOT, "include", "sbv.h"))
    for p in problems:
        print(p)
    print(f"{seen} cgo call sites checked against {len(protos)} prototypes of include/sbv.h: {'clean' if not problems else str(len(problems)) + ' problem(s)'}")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
