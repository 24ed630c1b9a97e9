"""No `julia` binary exists in the build container or on the GPU box, so the Julia files of this repo (the `ccall` glue, the closure
tracer, the parity kit) have never met a parser.  This is NOT a parser -- it is the cheapest class of error caught without one: every
block opener (`function`, `if`, `for`, `while`, `struct`, `module`, `let`, `begin`, `try`, `do`, `macro`, `quote`, `abstract type`) has its
`end`, every bracket its partner, strings and comments closed, per file; `end` inside brackets is indexing and `for` / `if` inside brackets
is a comprehension / generator, not a block."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "advancedmh.jl_amd", "julia", "*.jl")) + glob.glob(os.path.join(ROOT, "tests", "julia", "*.jl")))
OPENERS = {"function", "if", "for", "while", "struct", "module", "baremodule", "let", "begin", "try", "do", "macro", "quote"}


def strip(src):
    """remove comments and the contents of string / char literals (keeping the quotes), honouring triple quotes and escapes"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith('"""', i):
            j = src.find('"""', i + 3)
            assert j >= 0, "unterminated triple-quoted string at offset %d" % i
            out.append('""')
            i = j + 3
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            assert j < n, "unterminated string at offset %d" % i
            out.append('""')
            i = j + 1
        elif c == "#" and src.startswith("#=", i):
            j = src.find("=#", i + 2)
            assert j >= 0, "unterminated block comment"
            i = j + 2
        elif c == "#":
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif c == "'" and re.match(r"'(\\.|[^\\'])'", src[i:i + 4]):          # a character literal, not a transpose
            m = re.match(r"'(\\.|[^\\'])'", src[i:i + 4])
            out.append("' '")
            i += m.end()
        else:
            out.append(c)
            i += 1
    return "".join(out)


def balance(path):
    src = strip(open(path).read())
    depth, stack, line = 0, [], 1
    brackets = []
    for m in re.finditer(r"\n|[()\[\]{}]|\babstract\s+type\b|\bprimitive\s+type\b|\bmutable\s+struct\b|(?<![\w.:@$])[A-Za-z_]\w*(?![\w!])", src):
        t = m.group(0)
        if t == "\n":
            line += 1
        elif t in "([{":
            brackets.append((t, line))
        elif t in ")]}":
            assert brackets, "%s:%d: unmatched %s" % (path, line, t)
            o, ol = brackets.pop()
            assert "([{".index(o) == ")]}".index(t), "%s:%d: %s closes %s of line %d" % (path, line, t, o, ol)
        elif brackets:
            continue                                         # keywords inside brackets: comprehension `for` / `if`, indexing `end`
        elif t.startswith(("abstract", "primitive")) or t.startswith("mutable"):
            stack.append((t.split()[0], line))
        elif t in OPENERS:
            # `function` used as a value / short-form keywords do not occur in this repo's files; a symbol :for is excluded by the lookbehind
            stack.append((t, line))
        elif t == "end":
            assert stack, "%s:%d: `end` without an opener" % (path, line)
            stack.pop()
    assert not brackets, "%s: unclosed %s of line %d" % (path, brackets[-1][0], brackets[-1][1])
    assert not stack, "%s: `%s` of line %d has no `end`" % (path, stack[-1][0], stack[-1][1])
    return True


def test_julia_files_are_block_and_bracket_balanced():
    assert len(FILES) >= 5, FILES
    for f in FILES:
        assert balance(f), f


def test_the_checker_catches_what_it_claims(tmp_path):
    good = "module M\nfunction f(x)\n    y = [i for i in 1:3 if i > 1]\n    x > 0 ? y[end] : 0   # if end\nend\nstruct S; a::Int; end\nend\n"
    p = tmp_path / "g.jl"
    p.write_text(good)
    assert balance(str(p))
    for bad in (good.replace("    x > 0", "    if x > 0\n    x > 0"), good.replace("y[end]", "y[end"), good + "end\n"):
        p.write_text(bad)
        try:
            balance(str(p))
        except AssertionError:
            continue
        raise AssertionError("not caught: %r" % bad)
