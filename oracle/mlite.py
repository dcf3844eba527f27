"""TEST INFRASTRUCTURE — a small interpreter for the subset of MATLAB the reference's path is written in, so that the
reference's OWN source files (`/root/reference/GPz/GPz.m`, `getPHI.m`, `inv_logdet.m`, `Dxy.m`, `getPrior.m`, `fixPsi.m`,
`predictDiag.m`, `predictCov.m`) can be executed in this container, where neither MATLAB nor Octave exists.

What this is for.  `oracle/gpz_oracle.py` is a statement-level RESTATEMENT of those files; every other pin (finite differences, 50-digit
arithmetic, quadrature) checks the mathematics.  None of them checks that the restatement follows the reference's text — its index
conventions, operator precedence, loop order, which columns `iSigma_w(:,1:m,i)` takes.  Running the text itself does:
`oracle/run_reference.py` executes the .m files with this interpreter on seeded inputs and writes the outputs to
`tests/golden/ref_*.npz`; the CPU suite compares the oracle (and the GPU suite the HIP path) with those vectors, and re-executes the
files whenever `/root/reference` is present.  Nothing of the reference's text is stored in the repository: the files are read where
they lie at run time.

What this is not.  It is not MATLAB: it implements the constructs and builtins these files use (listed in `BUILTINS` and in
`Parser`), with MATLAB's documented semantics — column-major arrays, 1-based / logical / `end` indexing, `sum` along the first
non-singleton dimension, `'` as conjugate transpose, `A/B = A*inv(B)` by a solve, `svd(X,'econ')`, `eps(x)` as the spacing of x — on
NumPy float64.  A disagreement between a run of this interpreter and the oracle points at one of the two; agreement on every case
(objective, gradient, all outputs, all six methods, with and without input noise and missing values) is evidence that neither
misreads the reference.  Only tests/ and oracle/run_reference.py import it; the product never does.
"""
import math
import os
import re

import numpy as np

REF_DIR = "/root/reference/GPz"


class MError(Exception):
    pass


# ------------------------------------------------------------------------------------------------ lexer
TOK = re.compile(r"""
    (?P<ws>[ \t]+)
  | (?P<cont>\.\.\.[^\n]*\n)
  | (?P<comment>%[^\n]*)
  | (?P<num>(\d+(\.(?![*/\\^'])\d*)?|\.\d+)([eE][+-]?\d+)?)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>\.\*|\./|\.\\|\.\^|\.'|==|~=|<=|>=|&&|\|\||[-+*/\\^'<>=&|~(),;:\[\]{}@.\n])
""", re.X)
KEYWORDS = {"function", "end", "if", "elseif", "else", "for", "while", "switch", "case", "otherwise", "return", "break", "continue",
            "global"}


def lex(src):
    """-> list of (kind, text, space_before).  A quote is a transpose after an operand (no string can start there), else a string."""
    out = []
    pos, space = 0, False
    depth = []                                   # bracket stack: '(' or '['
    while pos < len(src):
        ch = src[pos]
        if ch == "'":
            prev = out[-1] if out else None
            operand = prev is not None and (prev[0] in ("num", "id", "str") or prev[1] in (")", "]", "}", "'", ".'")) and \
                not (prev[0] == "id" and prev[1] in KEYWORDS and prev[1] != "end")
            if operand and not (space and depth and depth[-1] == "["):
                out.append(("op", "'", space)); pos += 1; space = False
                continue
            end = pos + 1
            buf = []
            while True:
                if end >= len(src):
                    raise MError("unterminated string")
                if src[end] == "'":
                    if end + 1 < len(src) and src[end + 1] == "'":
                        buf.append("'"); end += 2
                        continue
                    break
                buf.append(src[end]); end += 1
            out.append(("str", "".join(buf), space)); pos = end + 1; space = False
            continue
        m = TOK.match(src, pos)
        if not m:
            raise MError("cannot tokenise at %r" % src[pos:pos + 20])
        pos = m.end()
        kind = m.lastgroup
        if kind in ("ws", "cont"):
            space = True
            continue
        if kind == "comment":
            continue
        text = m.group(kind)
        if text in ("(", "[", "{"):
            depth.append("[" if text == "{" else text)
        elif text in (")", "]", "}") and depth:
            depth.pop()
        out.append((kind, text, space))
        space = False
    out.append(("op", "\n", False))
    out.append(("eof", "", False))
    return out


# ------------------------------------------------------------------------------------------------ parser
class Parser:
    """Statements: function, if / elseif / else, for, while, switch / case / otherwise, global, return, break, continue, assignment
    (`x = e`, `x(i,j) = e`, `x(i) = []`, `[a,~,c] = f(...)`), expression.  Expressions by MATLAB's precedence:
    || && | & comparison : + - * / \\ .* ./ unary ^ .^ postfix(' .' (...) .field)."""

    def __init__(self, toks):
        self.t = toks
        self.i = 0
        self.in_matrix = 0
        self.in_index = 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, text):
        if self.peek()[1] == text and self.peek()[0] in ("op", "id"):
            return self.next()
        return None

    def expect(self, text):
        tok = self.next()
        if tok[1] != text:
            raise MError("expected %r, got %r" % (text, tok[1]))
        return tok

    def skip_newlines(self):
        while self.peek()[1] in ("\n", ";", ","):
            self.next()

    # ---- statements
    def parse_file(self):
        funcs = {}
        self.skip_newlines()
        while self.peek()[0] != "eof":
            f = self.parse_function()
            funcs.setdefault(f["name"], f)
            self.skip_newlines()
        return funcs

    def parse_function(self):
        self.expect("function")
        outs = []
        # forms: function name(...), function o = name(...), function [a,b] = name(...)
        if self.peek()[1] == "[":
            self.next()
            while self.peek()[1] != "]":
                if self.peek()[1] == ",":
                    self.next()
                    continue
                outs.append(self.next()[1])
            self.expect("]")
            self.expect("=")
            name = self.next()[1]
        else:
            name = self.next()[1]
            if self.peek()[1] == "=":
                self.next()
                outs = [name]
                name = self.next()[1]
        args = []
        if self.accept("("):
            while self.peek()[1] != ")":
                if self.peek()[1] == ",":
                    self.next()
                    continue
                args.append(self.next()[1])
            self.expect(")")
        body = self.parse_block(("end", "function"))
        if self.peek()[1] == "end":
            self.next()
        return {"name": name, "args": args, "outs": outs, "body": body}

    def parse_block(self, stops):
        body = []
        while True:
            self.skip_newlines()
            tok = self.peek()
            if tok[0] == "eof" or (tok[0] == "id" and tok[1] in stops):
                return body
            body.append(self.parse_statement())

    def parse_statement(self):
        tok = self.peek()
        if tok[0] == "id" and tok[1] in KEYWORDS:
            kw = tok[1]
            if kw == "if":
                self.next()
                clauses = []
                cond = self.parse_expr()
                blk = self.parse_block(("elseif", "else", "end"))
                clauses.append((cond, blk))
                other = None
                while True:
                    if self.accept("elseif"):
                        cond = self.parse_expr()
                        clauses.append((cond, self.parse_block(("elseif", "else", "end"))))
                    elif self.accept("else"):
                        other = self.parse_block(("end",))
                    else:
                        break
                self.expect("end")
                return ("if", clauses, other)
            if kw == "for":
                self.next()
                paren = self.accept("(")
                var = self.next()[1]
                self.expect("=")
                rng = self.parse_expr()
                if paren:
                    self.expect(")")
                body = self.parse_block(("end",))
                self.expect("end")
                return ("for", var, rng, body)
            if kw == "while":
                self.next()
                cond = self.parse_expr()
                body = self.parse_block(("end",))
                self.expect("end")
                return ("while", cond, body)
            if kw == "switch":
                self.next()
                subj = self.parse_expr()
                self.skip_newlines()
                cases, other = [], None
                while True:
                    if self.accept("case"):
                        val = self.parse_expr()
                        cases.append((val, self.parse_block(("case", "otherwise", "end"))))
                    elif self.accept("otherwise"):
                        other = self.parse_block(("end",))
                    else:
                        break
                self.expect("end")
                return ("switch", subj, cases, other)
            if kw == "global":
                self.next()
                names = []
                while self.peek()[0] == "id" and self.peek()[1] not in KEYWORDS:
                    names.append(self.next()[1])
                return ("global", names)
            if kw in ("return", "break", "continue"):
                self.next()
                return (kw,)
            raise MError("unexpected keyword " + kw)
        if tok[0] == "id" and tok[1] == "clear" and self.peek(1)[1] == "global":    # command form `clear global` (train.m:3)
            self.next(); self.next()
            return ("clearglobal",)
        # multi-assignment  [a,b,~] = f(...)
        if tok[1] == "[":
            save = self.i
            try:
                lhs = self.parse_lhs_list()
                if self.peek()[1] == "=" and self.peek(1)[1] != "=":
                    self.next()
                    rhs = self.parse_expr()
                    return ("massign", lhs, rhs)
            except MError:
                pass
            self.i = save
        expr = self.parse_expr()
        if self.peek()[1] == "=":
            self.next()
            rhs = self.parse_expr()
            return ("assign", expr, rhs)
        return ("expr", expr)

    def parse_lhs_list(self):
        self.expect("[")
        out = []
        while self.peek()[1] != "]":
            if self.peek()[1] == ",":
                self.next()
                continue
            if self.peek()[1] == "~":
                self.next()
                out.append(None)
                continue
            self.in_matrix += 1
            saved = self.in_matrix
            self.in_matrix = 0
            out.append(self.parse_postfix())
            self.in_matrix = saved - 1
        self.expect("]")
        return out

    # ---- expressions
    def parse_expr(self):
        return self.parse_binary(0)

    LEVELS = [("||",), ("&&",), ("|",), ("&",), ("==", "~=", "<", "<=", ">", ">=")]

    def parse_binary(self, lvl):
        if lvl == len(self.LEVELS):
            return self.parse_range()
        left = self.parse_binary(lvl + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[lvl] and not self.matrix_break():
            op = self.next()[1]
            right = self.parse_binary(lvl + 1)
            left = ("bin", op, left, right)
        return left

    def matrix_break(self):
        """inside [ ]: `a -b` starts a new element, `a - b` and `a-b` do not."""
        if not self.in_matrix:
            return False
        tok, nxt = self.peek(), self.peek(1)
        return tok[2] and tok[1] in ("+", "-") and not nxt[2]

    def parse_range(self):
        left = self.parse_additive()
        if self.peek()[1] == ":" and not (self.in_index and self.peek(1)[1] in (",", ")")):
            self.next()
            mid = self.parse_additive()
            if self.peek()[1] == ":":
                self.next()
                hi = self.parse_additive()
                return ("range", left, mid, hi)
            return ("range", left, None, mid)
        return left

    def parse_additive(self):
        left = self.parse_mul()
        while self.peek()[0] == "op" and self.peek()[1] in ("+", "-") and not self.matrix_break():
            op = self.next()[1]
            left = ("bin", op, left, self.parse_mul())
        return left

    def parse_mul(self):
        left = self.parse_unary()
        while self.peek()[0] == "op" and self.peek()[1] in ("*", "/", "\\", ".*", "./", ".\\"):
            op = self.next()[1]
            left = ("bin", op, left, self.parse_unary())
        return left

    def parse_unary(self):
        if self.peek()[0] == "op" and self.peek()[1] in ("-", "+", "~"):
            op = self.next()[1]
            return ("un", op, self.parse_unary())
        return self.parse_power()

    def parse_power(self):
        base = self.parse_postfix()
        while self.peek()[0] == "op" and self.peek()[1] in ("^", ".^"):
            op = self.next()[1]
            if self.peek()[1] in ("-", "+", "~"):                 # 2^-1, x.^-2
                u = self.next()[1]
                expo = ("un", u, self.parse_postfix())
            else:
                expo = self.parse_postfix()
            base = ("bin", op, base, expo)
        return base

    def parse_postfix(self):
        node = self.parse_primary()
        while True:
            tok = self.peek()
            if tok[1] == "(" and not (self.in_matrix and tok[2]):
                self.next()
                args = self.parse_args()
                node = ("index", node, args)
            elif tok[1] == "{" and not (self.in_matrix and tok[2]):
                self.next()
                args = self.parse_args("}")
                node = ("cellindex", node, args)
            elif tok[1] == "'" and tok[0] == "op":
                self.next()
                node = ("un", "'", node)
            elif tok[1] == ".'":
                self.next()
                node = ("un", ".'", node)
            elif tok[1] == "." and self.peek(1)[0] == "id" and not tok[2]:
                self.next()
                node = ("field", node, self.next()[1])
            else:
                return node

    def parse_args(self, close=")"):
        args = []
        saved_m, self.in_matrix = self.in_matrix, 0
        self.in_index += 1
        while self.peek()[1] != close:
            if self.peek()[1] == ",":
                self.next()
                continue
            if self.peek()[1] == ":" and self.peek(1)[1] in (",", ")", "}"):
                self.next()
                args.append(("colon",))
            else:
                args.append(self.parse_expr())
        self.expect(close)
        self.in_index -= 1
        self.in_matrix = saved_m
        return args

    def parse_primary(self):
        tok = self.next()
        if tok[0] == "num":
            return ("num", float(tok[1]))
        if tok[0] == "str":
            return ("str", tok[1])
        if tok[0] == "id":
            if tok[1] == "end" and self.in_index:
                return ("end",)
            if tok[1] in KEYWORDS:
                raise MError("unexpected keyword %s in expression" % tok[1])
            return ("name", tok[1])
        if tok[1] == "(":
            saved_m, self.in_matrix = self.in_matrix, 0
            saved_i = self.in_index                                  # `end` stays legal in a parenthesised part of a subscript: A((end - 1) / 2)
            e = self.parse_expr()
            self.expect(")")
            self.in_matrix, self.in_index = saved_m, saved_i
            return ("paren", e)
        if tok[1] == "[":
            rows, row = [], []
            self.in_matrix += 1
            saved_i = self.in_index                                  # ... and in a bracketed one: A(2:end, [1 end])
            while True:
                t = self.peek()
                if t[1] == "]":
                    self.next()
                    break
                if t[1] in (";", "\n"):
                    self.next()
                    if row:
                        rows.append(row)
                        row = []
                    continue
                if t[1] == ",":
                    self.next()
                    continue
                row.append(self.parse_expr())
            if row:
                rows.append(row)
            self.in_matrix -= 1
            self.in_index = saved_i
            return ("matrix", rows)
        if tok[1] == "{":
            items = []
            self.in_matrix += 1
            saved_i, self.in_index = self.in_index, 0
            while self.peek()[1] != "}":
                if self.peek()[1] in (",", ";", "\n"):
                    self.next()
                    continue
                items.append(self.parse_expr())
            self.next()
            self.in_matrix -= 1
            self.in_index = saved_i
            return ("cell", items)
        if tok[1] == "@":
            if self.peek()[1] == "(":                                # anonymous function: @(a,b,varargin) expression
                self.next()
                names = []
                while self.peek()[1] != ")":
                    t = self.next()
                    if t[1] != ",":
                        names.append(t[1])
                self.next()
                saved_m, saved_i = self.in_matrix, self.in_index
                self.in_matrix = self.in_index = 0
                body = self.parse_expr()
                self.in_matrix, self.in_index = saved_m, saved_i
                return ("lambda", names, body)
            return ("handle", self.next()[1])
        raise MError("unexpected token %r" % (tok[1],))


# ------------------------------------------------------------------------------------------------ values
class Struct:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def is_callable(v):
    return callable(v) or (isinstance(v, tuple) and len(v) > 0 and v[0] in ("handle", "closure"))


def mat(x):
    """every numeric value is an ndarray with at least two dimensions"""
    a = np.asarray(x)
    if a.dtype != bool and a.dtype.kind != "c":
        a = a.astype(np.float64)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(1, -1)
    return a


def trim(a):
    """drop trailing singleton dimensions beyond the second"""
    while a.ndim > 2 and a.shape[-1] == 1:
        a = a.reshape(a.shape[:-1])
    return a


def num(a):
    return a.astype(np.float64) if a.dtype == bool else a


def _sqrt(x):
    """sqrt of a negative real is complex, as in MATLAB (polyinterp.m:49-51 tests the result with isreal)"""
    if np.iscomplexobj(x) or np.any(x < 0):
        return np.sqrt(x.astype(np.complex128))
    return np.sqrt(x)


def _log(x):
    """log of a negative real is complex (log(-1) = 0 + 3.1416i), log(0) = -Inf"""
    if np.iscomplexobj(x) or np.any(x < 0):
        return np.log(x.astype(np.complex128))
    return np.log(x)


def _mod(a, b):
    """mod(a, 0) = a; otherwise the result has the sign of b (floored division)"""
    a, b = np.broadcast_arrays(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.mod(a, np.where(b == 0, 1.0, b))
    return np.where(b == 0, a, r)


def _nthroot(x, n):
    """real n-th root: negative x needs an odd integer n (nthroot(-27, 3) = -3)"""
    x, n = np.broadcast_arrays(np.asarray(x, dtype=np.float64), np.asarray(n, dtype=np.float64))
    if np.any((x < 0) & (np.mod(n, 2) != 1)):
        raise MError("nthroot: if x is negative, n must be an odd integer")
    with np.errstate(invalid="ignore"):
        return np.sign(x) * np.power(np.abs(x), 1.0 / n)


def _round_half_away(x):
    """round / integer conversion: ties away from zero (round(2.5) = 3, round(-2.5) = -3), unlike NumPy's ties-to-even"""
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def _struct(ip, a, n):
    return Struct(**{a[i]: a[i + 1] for i in range(0, len(a), 2)})


def _linsolve(ip, a, n):
    """linsolve(A,b): LU with partial pivoting (square; a singular matrix gives a warning and Inf, not an error), QR otherwise"""
    A, b = num(mat(a[0])), num(mat(a[1]))
    if A.shape[0] == A.shape[1]:
        try:
            x = np.linalg.solve(A, b)
        except np.linalg.LinAlgError:
            x = np.full((A.shape[1], b.shape[1]), np.inf)
    else:
        x = np.linalg.lstsq(A, b, rcond=None)[0]
    return [x, mat(0.0)]


def _roots(ip, a, n):
    c = num(mat(a[0])).reshape(-1)
    c = c[np.argmax(c != 0):] if np.any(c != 0) else c[:0]
    r = np.roots(c) if c.size > 1 else np.zeros(0)
    if np.all(np.imag(r) == 0):
        r = np.real(r)
    return r.reshape(-1, 1)


def scalar(v):
    if isinstance(v, str):
        raise MError("string where a number is expected")
    a = mat(v)
    if a.size != 1:
        raise MError("scalar expected, got shape %s" % (a.shape,))
    return float(a.reshape(-1)[0])


def truth(v):
    if isinstance(v, str):
        return len(v) > 0
    a = mat(v)
    return a.size > 0 and bool(np.all(a != 0))


class Return(Exception):
    pass


class Break(Exception):
    pass


class Continue(Exception):
    pass


class Colon:
    pass


# ------------------------------------------------------------------------------------------------ interpreter
class Interp:
    def __init__(self, ref_dir=REF_DIR):
        self.ref_dir = ref_dir                                       # one directory, or a list searched in order (the MATLAB path)
        self.funcs = {}
        self.globals = {}
        self.calls = 0
        self.extern = {}

    # ---- functions from the reference tree
    def load(self, name):
        if name in self.funcs:
            return self.funcs[name]
        dirs = [self.ref_dir] if isinstance(self.ref_dir, str) else list(self.ref_dir)
        path = next((q for q in (os.path.join(dd, name + ".m") for dd in dirs) if os.path.exists(q)), None)
        if path is None:
            return None
        with open(path) as fh:
            src = fh.read()
        funcs = Parser(lex(src)).parse_file()
        for fname, f in funcs.items():          # the first function is the file's; the rest are its local functions
            self.funcs.setdefault(fname, f)
        return self.funcs.get(name)

    def call(self, name, args, nargout=1):
        f = self.load(name)
        if f is None:
            raise MError("unknown function " + name)
        self.calls += 1
        scope = {"__globals__": set(), "nargout": mat(float(nargout)), "nargin": mat(float(len(args)))}
        names = f["args"]
        if names and names[-1] == "varargin":
            scope["varargin"] = list(args[len(names) - 1:])
            names = names[:-1]
        for a, v in zip(names, args):
            scope[a] = v
        try:
            self.run_block(f["body"], scope)
        except Return:
            pass
        outs = []
        for o in f["outs"][:max(nargout, 1)]:
            if o not in scope:
                if len(outs) >= nargout:
                    break
                raise MError("output %s of %s not assigned" % (o, name))
            outs.append(scope[o])
        return outs

    # ---- statements
    def get(self, scope, name):
        if name in scope["__globals__"]:
            return self.globals.get(name, np.zeros((0, 0)))
        return scope[name]

    def put(self, scope, name, val):
        if name in scope["__globals__"]:
            self.globals[name] = val
        else:
            scope[name] = val

    def has(self, scope, name):
        return name in scope or name in scope["__globals__"]

    def run_block(self, body, scope):
        for st in body:
            self.run(st, scope)

    def run(self, st, scope):
        kind = st[0]
        if kind == "expr":
            node = st[1]
            if node[0] == "name" and not self.has(scope, node[1]):     # command-form call of a user function
                self.call_any(node[1], [], 0, scope)
            elif node[0] == "index" and node[1][0] == "name" and not self.has(scope, node[1][1]):
                self.call_any(node[1][1], self.ev_args(node[2], scope), 0, scope)
            elif node[0] == "index" and node[1][0] == "name" and is_callable(self.get(scope, node[1][1])):
                self.call_handle(self.get(scope, node[1][1]), self.ev_args(node[2], scope), 0, scope)
            else:
                self.ev(node, scope)
        elif kind == "assign":
            self.assign(st[1], self.ev(st[2], scope), scope, st[2])
        elif kind == "massign":
            lhs, rhs = st[1], st[2]
            if rhs[0] == "index" and rhs[1][0] == "name" and not self.has(scope, rhs[1][1]):
                args = self.ev_args(rhs[2], scope)
                vals = self.call_any(rhs[1][1], args, len(lhs), scope)
            elif rhs[0] == "name" and not self.has(scope, rhs[1]):
                vals = self.call_any(rhs[1], [], len(lhs), scope)
            elif rhs[0] == "index" and rhs[1][0] == "name" and is_callable(self.get(scope, rhs[1][1])):
                vals = self.call_handle(self.get(scope, rhs[1][1]), self.ev_args(rhs[2], scope), len(lhs), scope)
            elif rhs[0] == "index" and self.dotted(rhs[1]) == "internal.stats.parseArgs":
                vals = parse_args_builtin(self.ev_args(rhs[2], scope))
            else:
                vals = [self.ev(rhs, scope)]
            if len(vals) < len([x for x in lhs if x is not None]) and len(vals) < len(lhs):
                raise MError("too many outputs requested")
            for tgt, v in zip(lhs, vals):
                if tgt is not None:
                    self.assign(tgt, v, scope, None)
        elif kind == "if":
            for cond, blk in st[1]:
                if truth(self.ev(cond, scope)):
                    self.run_block(blk, scope)
                    return
            if st[2] is not None:
                self.run_block(st[2], scope)
        elif kind == "for":
            rng = self.ev(st[2], scope)
            rng = mat(rng)
            cols = rng.reshape(rng.shape[0], -1, order="F")
            for c in range(cols.shape[1]):
                self.put(scope, st[1], cols[:, c:c + 1] if cols.shape[0] > 1 else cols[:, c:c + 1].reshape(1, 1))
                try:
                    self.run_block(st[3], scope)
                except Break:
                    break
                except Continue:
                    continue
        elif kind == "while":
            while truth(self.ev(st[1], scope)):
                try:
                    self.run_block(st[2], scope)
                except Break:
                    break
                except Continue:
                    continue
        elif kind == "switch":
            subj = self.ev(st[1], scope)
            for val, blk in st[2]:
                v = self.ev(val, scope)
                hit = (subj == v) if isinstance(subj, str) or isinstance(v, str) else bool(scalar(subj) == scalar(v))
                if hit:
                    self.run_block(blk, scope)
                    return
            if st[3] is not None:
                self.run_block(st[3], scope)
        elif kind == "global":
            for n in st[1]:
                scope["__globals__"].add(n)
                scope.pop(n, None)
        elif kind == "clearglobal":
            self.globals.clear()
        elif kind == "return":
            raise Return()
        elif kind == "break":
            raise Break()
        elif kind == "continue":
            raise Continue()
        else:
            raise MError("statement " + kind)

    # ---- assignment
    def assign(self, target, val, scope, rhs_node):
        if target[0] == "name":
            self.put(scope, target[1], val)
            return
        if target[0] == "field":
            base = target[1]                                          # structs are values: the holder gets a modified copy
            if base[0] == "name":
                obj = self.get(scope, base[1]) if self.has(scope, base[1]) else Struct()
            elif base[0] == "field":
                try:
                    obj = self.ev(base, scope)
                except (AttributeError, KeyError):
                    obj = Struct()
            else:
                raise MError("field assignment on an indexed struct")
            obj = Struct(**obj.__dict__) if isinstance(obj, Struct) else Struct()
            setattr(obj, target[2], val)
            self.assign(base, obj, scope, None)
            return
        if target[0] == "index" and target[1][0] == "name":
            name = target[1][1]
            if self.has(scope, name) and isinstance(self.get(scope, name), str) and isinstance(val, str) and len(val) == 1:   # method(2) = 'L'
                chars = list(self.get(scope, name))
                for q in self.to_index(self.subscripts(target[2], np.zeros((1, len(chars))), scope)[-1], len(chars)):
                    chars[int(q)] = val
                self.put(scope, name, "".join(chars))
                return
            cur = mat(self.get(scope, name)) if self.has(scope, name) else np.zeros((0, 0))
            subs = self.subscripts(target[2], cur, scope)
            is_delete = rhs_node is not None and rhs_node == ("matrix", [])
            self.put(scope, name, self.delete(cur, subs) if is_delete else self.store(cur, subs, val))
            return
        if target[0] == "index" and target[1][0] == "field" and target[1][1][0] == "name":      # s.f(subs) = v
            fld = target[1]
            obj = self.get(scope, fld[1][1]) if self.has(scope, fld[1][1]) else Struct()
            cur = mat(getattr(obj, fld[2])) if hasattr(obj, fld[2]) else np.zeros((0, 0))
            self.assign(fld, self.store(cur, self.subscripts(target[2], cur, scope), val), scope, None)
            return
        if target[0] == "cellindex" and target[1][0] == "name":                                # c{i} = v
            name = target[1][1]
            cell = list(self.get(scope, name)) if self.has(scope, name) else []
            idx = int(scalar(self.ev(target[2][0], scope))) - 1
            while len(cell) <= idx:
                cell.append(np.zeros((0, 0)))
            cell[idx] = val
            self.put(scope, name, cell)
            return
        raise MError("cannot assign to %r" % (target[0],))

    def subscripts(self, nodes, arr, scope):
        subs = []
        n = len(nodes)
        for pos, nd in enumerate(nodes):
            if nd[0] == "colon":
                subs.append(Colon())
                continue
            if n == 1:
                extent = arr.size
            elif pos == n - 1:
                extent = int(np.prod(arr.shape[pos:])) if pos < arr.ndim else 1
            else:
                extent = arr.shape[pos] if pos < arr.ndim else 1
            scope["__end__"] = scope.get("__end__", []) + [extent]
            try:
                subs.append(self.ev(nd, scope))
            finally:
                scope["__end__"].pop()
        return subs

    @staticmethod
    def to_index(s, extent):
        """-> 0-based integer index vector for one subscript"""
        if isinstance(s, Colon):
            return np.arange(extent)
        a = mat(s)
        if a.dtype == bool:
            flat = a.reshape(-1, order="F")
            if flat.size > extent and flat[extent:].any():
                raise MError("logical index out of range")
            return np.flatnonzero(flat[:extent] if flat.size > extent else flat)
        idx = a.reshape(-1, order="F")
        if np.any(idx != np.round(idx)) or np.any(idx < 1):
            raise MError("subscript indices must be positive integers")
        return idx.astype(np.int64) - 1

    def load_index(self, arr, subs):
        out = self.load_index_raw(arr, subs)
        if np.iscomplexobj(out) and not np.any(np.imag(out)):          # a subscripted reference with no imaginary part is real
            out = np.real(out).astype(np.float64)
        return out

    def load_index_raw(self, arr, subs):
        arr = mat(arr)
        if len(subs) == 1:
            s = subs[0]
            flat = arr.reshape(-1, order="F")
            if isinstance(s, Colon):
                return flat.reshape(-1, 1)
            idx = self.to_index(s, flat.size)
            if idx.size and idx.max() >= flat.size:
                raise MError("index exceeds array bounds")
            out = flat[idx]
            sa = mat(s)
            if sa.dtype == bool:
                return out.reshape(1, -1) if (arr.ndim == 2 and arr.shape[0] == 1) else out.reshape(-1, 1)
            if arr.ndim == 2 and min(arr.shape) == 1 and min(sa.shape) == 1:      # vector indexed by a vector: orientation of arr
                return out.reshape(1, -1) if arr.shape[0] == 1 else out.reshape(-1, 1)
            return out.reshape(sa.shape, order="F")
        shape = list(arr.shape) + [1] * (len(subs) - arr.ndim)
        if len(subs) < arr.ndim:                                       # last subscript spans the remaining dimensions
            shape = shape[:len(subs) - 1] + [int(np.prod(shape[len(subs) - 1:]))]
        a = arr.reshape(shape, order="F")
        idx = [self.to_index(s, shape[q]) for q, s in enumerate(subs)]
        for q, ix in enumerate(idx):
            if ix.size and ix.max() >= shape[q]:
                raise MError("index exceeds array bounds")
        return trim(a[np.ix_(*idx)])

    def store(self, arr, subs, val):
        v = mat(val) if not isinstance(val, str) else val
        if isinstance(v, str):
            raise MError("string assignment into an array")
        if arr.dtype == bool and v.dtype != bool:
            arr = arr.astype(np.float64)
        elif arr.dtype != bool and v.dtype == bool:
            v = v.astype(np.float64)
        if len(subs) == 1:
            flat = arr.reshape(-1, order="F").copy()
            idx = self.to_index(subs[0], flat.size)
            if idx.size and idx.max() >= flat.size:                    # growth of a vector
                if arr.size and min(arr.shape) != 1:
                    raise MError("linear-index growth of a matrix")
                grown = np.zeros(idx.max() + 1, dtype=flat.dtype)
                grown[:flat.size] = flat
                flat = grown
                shape = (1, flat.size) if (arr.size == 0 or arr.shape[0] == 1) and not (arr.size and arr.shape[1] == 1 and arr.shape[0] > 1) else (flat.size, 1)
            else:
                shape = arr.shape
            vv = v.reshape(-1, order="F")
            if vv.size == 1:
                flat[idx] = vv[0]
            else:
                if vv.size != idx.size:
                    raise MError("assignment size mismatch (%d vs %d)" % (vv.size, idx.size))
                flat[idx] = vv
            return flat.reshape(shape, order="F")
        shape = list(arr.shape) + [1] * (len(subs) - arr.ndim)
        if len(subs) < arr.ndim:
            raise MError("assignment with fewer subscripts than dimensions")
        idx = []
        for q, s in enumerate(subs):
            ix = self.to_index(s, shape[q])
            if ix.size and ix.max() >= shape[q]:
                shape[q] = int(ix.max()) + 1                           # growth
            idx.append(ix)
        out = np.zeros(shape, dtype=arr.dtype if arr.size else v.dtype)
        if arr.size:
            src = arr.reshape(list(arr.shape) + [1] * (len(shape) - arr.ndim), order="F")
            out[tuple(slice(0, e) for e in src.shape)] = src
        tgt_shape = tuple(ix.size for ix in idx)
        if v.size == 1:
            out[np.ix_(*idx)] = v.reshape(-1)[0]
        else:
            vs = [e for e in v.shape if e != 1]
            ts = [e for e in tgt_shape if e != 1]
            if vs != ts:
                raise MError("assignment dimension mismatch: %s into %s" % (v.shape, tgt_shape))
            out[np.ix_(*idx)] = v.reshape(tgt_shape, order="F")
        return trim(out)

    def delete(self, arr, subs):
        if len(subs) == 1:
            flat = arr.reshape(-1, order="F")
            keep = np.ones(flat.size, dtype=bool)
            keep[self.to_index(subs[0], flat.size)] = False
            out = flat[keep]
            return out.reshape(1, -1) if arr.shape[0] == 1 else out.reshape(-1, 1)
        full = [isinstance(s, Colon) for s in subs]
        if sum(not f for f in full) != 1:
            raise MError("deletion needs exactly one non-colon subscript")
        ax = full.index(False)
        keep = np.ones(arr.shape[ax], dtype=bool)
        keep[self.to_index(subs[ax], arr.shape[ax])] = False
        return np.compress(keep, arr, axis=ax)

    # ---- expressions
    def ev_arg(self, node, scope):
        if node[0] == "colon":
            return ":"
        return self.ev(node, scope)

    def ev_args(self, nodes, scope):
        """argument list of a call; `c{:}` expands to the cell's elements"""
        out = []
        for nd in nodes:
            if nd[0] == "cellindex" and len(nd[2]) == 1 and nd[2][0][0] == "colon":
                out.extend(self.ev(nd[1], scope))
            else:
                out.append(self.ev_arg(nd, scope))
        return out

    def call_handle(self, h, args, nargout, scope):
        """@name, @(args) expression, or a Python callable handed in by the driver (called as f(args, nargout) -> list)"""
        if callable(h):
            return list(h(args, nargout))
        if h[0] == "handle":
            return self.call_any(h[1], args, nargout, scope)
        _, names, body, captured = h
        inner = dict(captured)
        inner["__globals__"] = set()
        if names and names[-1] == "varargin":
            inner["varargin"] = list(args[len(names) - 1:])
            names = names[:-1]
        for a, v in zip(names, args):
            inner[a] = v
        if body[0] == "index" and body[1][0] == "name" and not self.has(inner, body[1][1]):
            return self.call_any(body[1][1], self.ev_args(body[2], inner), nargout, inner)
        if body[0] == "index" and body[1][0] == "name" and is_callable(self.get(inner, body[1][1])):
            return self.call_handle(self.get(inner, body[1][1]), self.ev_args(body[2], inner), nargout, inner)
        return [self.ev(body, inner)]

    def call_any(self, name, args, nargout, scope):
        if name in self.extern:                                      # functions the driver stands in for (compiled MEX files)
            return list(self.extern[name](args, nargout))
        if name in BUILTINS:
            BUILTINS_REACHED[name] = BUILTINS_REACHED.get(name, 0) + 1     # coverage list of tests/test_reference_run.py
            out = BUILTINS[name](self, args, nargout)
            return out if isinstance(out, list) else [out]
        return self.call(name, args, nargout)

    def ev(self, node, scope):
        kind = node[0]
        if kind == "num":
            return mat(node[1])
        if kind == "str":
            return node[1]
        if kind == "paren":
            return self.ev(node[1], scope)
        if kind == "name":
            name = node[1]
            if self.has(scope, name):
                return self.get(scope, name)
            return self.call_any(name, [], 1, scope)[0]
        if kind == "end":
            return mat(float(scope["__end__"][-1]))
        if kind == "handle":
            return ("handle", node[1])
        if kind == "lambda":                                         # the workspace is captured by value when the handle is made
            return ("closure", node[1], node[2], {k: v for k, v in scope.items() if k not in ("__end__",)})
        if kind == "field":
            return getattr(self.ev(node[1], scope), node[2])
        if kind == "range":
            lo = scalar(self.ev(node[1], scope))
            hi = scalar(self.ev(node[3], scope))
            step = 1.0 if node[2] is None else scalar(self.ev(node[2], scope))
            nsteps = int(math.floor((hi - lo) / step + 1e-10)) + 1
            return mat(lo + step * np.arange(max(nsteps, 0))).reshape(1, -1)
        if kind == "matrix":
            return self.build_matrix(node[1], scope)
        if kind == "un":
            v = self.ev(node[2], scope)
            op = node[1]
            if op in ("'", ".'"):
                if isinstance(v, str):
                    return v
                a = mat(v)
                if a.ndim > 2:
                    raise MError("transpose of an N-D array")
                return a.T
            a = mat(v)
            if op == "-":
                return -num(a)
            if op == "+":
                return num(a)
            return ~(a != 0) if a.dtype != bool else ~a
        if kind == "bin":
            if node[1] in ("||", "&&"):                              # short-circuit: the right side may name variables that do not exist yet
                left = truth(self.ev(node[2], scope))
                if left == (node[1] == "||"):
                    return mat(left)
                return mat(truth(self.ev(node[3], scope)))
            return self.binop(node[1], self.ev(node[2], scope), self.ev(node[3], scope))
        if kind == "cell":
            return [self.ev(e, scope) for e in node[1]]
        if kind == "cellindex":
            cell = self.ev(node[1], scope)
            subs = self.subscripts(node[2], np.zeros((1, len(cell))), scope)
            idx = self.to_index(subs[-1], len(cell))
            if idx.size != 1:
                raise MError("brace indexing with several elements outside an argument list")
            return cell[int(idx[0])]
        if kind == "index":
            base = node[1]
            if base[0] == "name" and not self.has(scope, base[1]):
                args = self.ev_args(node[2], scope)
                return self.call_any(base[1], args, 1, scope)[0]
            target = self.ev(base, scope)
            if isinstance(target, str):
                subs = self.subscripts(node[2], mat(np.zeros((1, len(target)))), scope)
                idx = self.to_index(subs[-1], len(target))
                return "".join(target[q] for q in idx)
            if is_callable(target):
                return self.call_handle(target, self.ev_args(node[2], scope), 1, scope)[0]
            arr = mat(target)
            return self.load_index(arr, self.subscripts(node[2], arr, scope))
        raise MError("expression " + kind)

    def build_matrix(self, rows, scope):
        if not rows:
            return np.zeros((0, 0))
        out_rows = []
        for row in rows:
            vals = [self.ev(e, scope) for e in row]
            if all(isinstance(v, str) for v in vals):
                out_rows.append("".join(vals))
                continue
            parts = [mat(v) for v in vals]
            parts = [p for p in parts if p.size > 0]
            if not parts:
                continue
            if any(p.dtype != bool for p in parts):
                parts = [num(p) for p in parts]
            out_rows.append(np.concatenate(parts, axis=1))
        if not out_rows:
            return np.zeros((0, 0))
        if all(isinstance(r, str) for r in out_rows):
            if len(out_rows) != 1:
                raise MError("multi-row char array")
            return out_rows[0]
        if any(r.dtype != bool for r in out_rows):
            out_rows = [num(r) for r in out_rows]
        return np.concatenate(out_rows, axis=0)

    def binop(self, op, a, b):
        if isinstance(a, str) or isinstance(b, str):
            if op in ("==", "~="):
                if isinstance(a, str) and isinstance(b, str) and (len(a) == len(b) or len(a) == 1 or len(b) == 1):
                    n = max(len(a), len(b))
                    aa = a * n if len(a) == 1 else a
                    bb = b * n if len(b) == 1 else b
                    r = np.array([[x == y for x, y in zip(aa, bb)]])
                    return r if op == "==" else ~r
            raise MError("operator %s on strings" % op)
        A, B = mat(a), mat(b)
        if op in ("&&", "||"):
            return mat(truth(A) and truth(B)) if op == "&&" else mat(truth(A) or truth(B))
        if op in ("==", "~=", "<", "<=", ">", ">=", "&", "|"):
            self.same_or_scalar(A, B, op)
            x, y = num(A), num(B)
            return {"==": x == y, "~=": x != y, "<": x < y, "<=": x <= y, ">": x > y, ">=": x >= y,
                    "&": (x != 0) & (y != 0), "|": (x != 0) | (y != 0)}[op]
        A, B = num(A), num(B)
        if op in ("+", "-", ".*", "./", ".\\", ".^"):
            self.same_or_scalar(A, B, op)
            if op == "+":
                return A + B
            if op == "-":
                return A - B
            if op == ".*":
                return A * B
            if op == "./":
                with np.errstate(divide="ignore", invalid="ignore"):
                    return A / B
            if op == ".\\":
                with np.errstate(divide="ignore", invalid="ignore"):
                    return B / A
            with np.errstate(divide="ignore", invalid="ignore"):
                return np.power(A, B)
        if A.ndim > 2 or B.ndim > 2:
            raise MError("matrix operator on an N-D array")
        if op == "*":
            if A.size == 1 or B.size == 1:
                return A * B
            if A.shape[1] != B.shape[0]:
                raise MError("inner matrix dimensions must agree: %s * %s" % (A.shape, B.shape))
            return A @ B
        if op == "/":                                                   # A/B = A*inv(B): x B = A
            if B.size == 1:
                return A / B
            if B.shape[0] != B.shape[1]:
                raise MError("mrdivide with a non-square matrix")
            return np.linalg.solve(B.T, A.T).T
        if op == "\\":
            if A.size == 1:
                return B / A
            if A.shape[0] != A.shape[1]:
                raise MError("mldivide with a non-square matrix")
            return np.linalg.solve(A, B)
        if op == "^":
            if A.size == 1 and B.size == 1:
                return np.power(A, B)
            if B.size == 1 and A.shape[0] == A.shape[1] and float(B) == int(float(B)):
                return np.linalg.matrix_power(A, int(float(B)))
            raise MError("matrix power")
        raise MError("operator " + op)

    @staticmethod
    def dotted(node):
        if node[0] == "name":
            return node[1]
        if node[0] == "field":
            base = Interp.dotted(node[1])
            return None if base is None else base + "." + node[2]
        return None

    @staticmethod
    def same_or_scalar(A, B, op):
        """implicit expansion did not exist when the reference was written: elementwise operands are equal-sized or scalar"""
        if A.size == 1 or B.size == 1 or trim(A).shape == trim(B).shape:
            return
        raise MError("matrix dimensions must agree for %s: %s vs %s" % (op, A.shape, B.shape))


# ------------------------------------------------------------------------------------------------ builtins
def _dims(args):
    if len(args) == 1:
        a = mat(args[0])
        if a.size == 1:
            n = int(scalar(a))
            return (n, n)
        return tuple(int(x) for x in a.reshape(-1))
    return tuple(int(scalar(a)) for a in args)


def _first_dim(a):
    for ax, e in enumerate(a.shape):
        if e != 1:
            return ax
    return 0


def _reduce(fn):
    def f(ip, args, nargout):
        a = num(mat(args[0]))
        if a.shape == (0, 0) and len(args) == 1:
            return mat(0.0 if fn is np.sum else 1.0)              # sum([]) is 0, prod([]) is 1
        ax = _first_dim(a) if len(args) == 1 else int(scalar(args[1])) - 1
        if ax >= a.ndim:
            return a
        return trim(fn(a, axis=ax, keepdims=True))
    return f


def _elementwise(fn):
    def f(ip, args, nargout):
        with np.errstate(divide="ignore", invalid="ignore"):
            return fn(num(mat(args[0])))
    return f


def _size(ip, args, nargout):
    v = args[0]
    shape = (1, len(v)) if isinstance(v, str) else mat(v).shape
    if len(args) == 2:
        q = int(scalar(args[1])) - 1
        return mat(float(shape[q] if q < len(shape) else 1))
    if nargout <= 1:
        return mat(np.array(shape, dtype=np.float64)).reshape(1, -1)
    out = [float(e) for e in shape[:nargout - 1]] + [float(np.prod(shape[nargout - 1:]))]
    return [mat(e) for e in out] + [mat(1.0)] * (nargout - len(out))


def _bsxfun(ip, args, nargout):
    h, A, B = args
    A, B = num(mat(A)), num(mat(B))
    nd = max(A.ndim, B.ndim)
    A = A.reshape(A.shape + (1,) * (nd - A.ndim))
    B = B.reshape(B.shape + (1,) * (nd - B.ndim))
    for ea, eb in zip(A.shape, B.shape):
        if ea != eb and ea != 1 and eb != 1:
            raise MError("bsxfun: non-singleton dimensions must match: %s vs %s" % (A.shape, B.shape))
    name = h[1]
    with np.errstate(divide="ignore", invalid="ignore"):
        if name == "times":
            return trim(A * B)
        if name == "minus":
            return trim(A - B)
        if name == "plus":
            return trim(A + B)
        if name == "rdivide":
            return trim(A / B)
        if name == "power":
            return trim(np.power(A, B))
        if name == "eq":
            return trim(A == B)
    raise MError("bsxfun handle " + name)


def _find(ip, args, nargout):
    a = mat(args[0])
    idx = np.flatnonzero(a.reshape(-1, order="F") != 0) + 1.0
    if len(args) > 1:
        idx = idx[:int(scalar(args[1]))]
    return idx.reshape(1, -1) if (a.ndim == 2 and a.shape[0] == 1 and a.shape[1] != 1) else idx.reshape(-1, 1)


def _diag(ip, args, nargout):
    a = num(mat(args[0]))
    if min(a.shape) == 1:
        return np.diag(a.reshape(-1))
    return np.diag(a).reshape(-1, 1).copy()


def _svd(ip, args, nargout):
    a = num(mat(args[0]))
    if nargout <= 1:
        return np.linalg.svd(a, compute_uv=False).reshape(-1, 1)
    econ = len(args) > 1
    U, s, Vh = np.linalg.svd(a, full_matrices=not econ)
    S = np.zeros((U.shape[1], Vh.shape[0]))
    S[:s.size, :s.size] = np.diag(s)
    return [U, S, Vh.T]


def _reshape(ip, args, nargout):
    """reshape(A, sz), reshape(A, m, n, ...), one dimension may be [] (computed from numel)"""
    a = mat(args[0])
    rest = args[1:]
    if len(rest) > 1 and any(mat(r).size == 0 for r in rest):
        known = [int(scalar(r)) for r in rest if mat(r).size]
        prod = int(np.prod(known)) if known else 1
        if sum(mat(r).size == 0 for r in rest) != 1 or prod == 0 or a.size % prod:
            raise MError("reshape: one [] placeholder, and the known dimensions must divide numel")
        dims = tuple(int(scalar(r)) if mat(r).size else a.size // prod for r in rest)
    else:
        dims = _dims(rest)
    return trim(a.reshape(dims, order="F"))


def _repmat(ip, args, nargout):
    a = mat(args[0])
    reps = _dims(args[1:])
    nd = max(a.ndim, len(reps))
    return trim(np.tile(a.reshape(a.shape + (1,) * (nd - a.ndim)), reps + (1,) * (nd - len(reps))))


def _norm(ip, args, nargout):
    a = num(mat(args[0]))
    if len(args) == 1:
        return mat(np.linalg.norm(a.reshape(-1)) if min(a.shape) == 1 else np.linalg.norm(a, 2))
    p = args[1]
    if isinstance(p, str):
        if p == "fro":
            return mat(np.linalg.norm(a, "fro"))
        raise MError("norm " + p)
    p = scalar(p)
    if min(a.shape) == 1:
        return mat(np.linalg.norm(a.reshape(-1), p))
    return mat(np.linalg.norm(a, p))


def _minmax(fn, argfn):
    def f(ip, args, nargout):
        # NaNs are ignored (the default 'omitnan' of max / min); a slice that is all NaN gives NaN, with index 1
        a = num(mat(args[0]))
        if len(args) >= 2 and mat(args[1]).size > 0:
            return (np.fmax if fn is np.maximum else np.fmin)(a, num(mat(args[1])))
        if a.size == 0:
            return a
        ax = _first_dim(a) if len(args) < 3 else int(scalar(args[2])) - 1
        big = fn is np.maximum
        nanmask = np.isnan(a)
        filled = np.where(nanmask, -np.inf if big else np.inf, a)
        vals = (np.max if big else np.min)(filled, axis=ax, keepdims=True)
        allnan = nanmask.all(axis=ax, keepdims=True)
        vals = np.where(allnan, np.nan, vals)
        if nargout >= 2:
            idx = argfn(filled, axis=ax, keepdims=True) + 1.0
            return [trim(vals), trim(np.where(allnan, 1.0, idx))]
        return trim(vals)
    return f


def _sort(ip, args, nargout):
    a = num(mat(args[0]))
    ax = _first_dim(a)
    desc = len(args) > 1 and isinstance(args[-1], str) and args[-1] == "descend"
    idx = np.argsort(-a if desc else a, axis=ax, kind="stable")
    vals = np.take_along_axis(a, idx, axis=ax)
    return [vals, idx + 1.0] if nargout >= 2 else vals


def _squeeze(ip, args, nargout):
    a = mat(args[0])
    if a.ndim <= 2:
        return a
    shape = [e for e in a.shape if e != 1]
    while len(shape) < 2:
        shape.append(1)
    return a.reshape(shape, order="F")


def _cumsum(ip, args, nargout):
    a = num(mat(args[0]))
    return np.cumsum(a, axis=_first_dim(a) if len(args) == 1 else int(scalar(args[1])) - 1)


def _mean(ip, args, nargout):
    a = num(mat(args[0]))
    ax = _first_dim(a) if len(args) == 1 else int(scalar(args[1])) - 1
    return trim(np.mean(a, axis=ax, keepdims=True))


def _var(ip, args, nargout):
    """var(A), var(A, w), var(A, w, dim): w = 0 (default) normalises by N - 1, w = 1 by N; vector weights are not implemented"""
    a = num(mat(args[0]))
    w = 0.0
    if len(args) >= 2 and mat(args[1]).size > 0:
        if mat(args[1]).size != 1:
            raise MError("var: weight vectors are not implemented")
        w = scalar(args[1])
    ax = _first_dim(a) if len(args) < 3 else int(scalar(args[2])) - 1
    return trim(np.var(a, axis=ax, ddof=1 if (w == 0.0 and a.shape[ax] > 1) else 0, keepdims=True))


def _eig(ip, args, nargout):
    """eig of a real symmetric matrix (pca.m:19): ascending eigenvalues; [V,D] = eig(A) with nargout 2"""
    A = num(mat(args[0]))
    if not np.allclose(A, A.T, rtol=1e-12, atol=1e-300):
        raise MError("eig: only symmetric input is implemented")
    w, V = np.linalg.eigh(0.5 * (A + A.T))
    return [V, np.diag(w)] if nargout >= 2 else [w.reshape(-1, 1)]


def _hist(ip, args, nargout):
    """n = hist(y, centers): counts per bin, bin edges midway between the centres, outer bins open-ended; row vector for vector y"""
    y = num(mat(args[0])).reshape(-1, order="F")
    c = num(mat(args[1])).reshape(-1, order="F") if len(args) > 1 else np.array([10.0])
    if c.size == 1:                                                  # hist(y, nbins): equally spaced bins between min(y) and max(y)
        nb = int(c[0])
        lo, hi = float(np.min(y)), float(np.max(y))
        if lo == hi:
            lo, hi = lo - nb / 2.0, hi + nb / 2.0
        width = (hi - lo) / nb
        c = lo + width * (np.arange(nb) + 0.5)
    edges = np.concatenate(([-np.inf], 0.5 * (c[1:] + c[:-1]), [np.inf]))
    counts = np.histogram(y, bins=edges)[0].astype(np.float64)
    # MATLAB puts a value that sits exactly on an edge into the upper bin, as np.histogram does (right-open bins)
    return [counts.reshape(1, -1), c.reshape(1, -1)] if nargout >= 2 else counts.reshape(1, -1)


def _sparse(a):
    """sparse(A), sparse(m,n) or sparse(i,j,v,m,n) (duplicates add up) - held dense"""
    if len(a) == 1:
        return num(mat(a[0]))
    if len(a) == 2:
        return np.zeros(_dims(a))
    i = mat(a[0]).reshape(-1, order="F").astype(np.int64) - 1
    j = mat(a[1]).reshape(-1, order="F").astype(np.int64) - 1
    v = num(mat(a[2])).reshape(-1, order="F")
    nn = max(i.size, j.size, v.size)
    i, j, v = (np.broadcast_to(x, (nn,)) if x.size == 1 else x for x in (i, j, v))
    shape = (int(scalar(a[3])), int(scalar(a[4]))) if len(a) >= 5 else (int(i.max()) + 1, int(j.max()) + 1)
    out = np.zeros(shape)
    np.add.at(out, (i, j), v)
    return out


def _setfield(ip, a, n):
    o = Struct(**a[0].__dict__) if isinstance(a[0], Struct) else Struct()
    setattr(o, a[1], a[2])
    return o


def _length(v):
    if isinstance(v, (str, list)):
        return len(v)
    a = mat(v)
    return 0 if a.size == 0 else max(a.shape)


BUILTINS_REACHED = {}      # name -> calls, filled while reference files execute (which builtins the pins must cover)

BUILTINS = {
    "fieldnames": lambda ip, a, n: [list(a[0].__dict__.keys())],
    "getfield": lambda ip, a, n: getattr(a[0], a[1]),
    "setfield": _setfield,
    "isfield": lambda ip, a, n: mat(isinstance(a[0], Struct) and isinstance(a[1], str) and hasattr(a[0], a[1])),
    "upper": lambda ip, a, n: a[0].upper() if isinstance(a[0], str) else a[0],
    "lower": lambda ip, a, n: a[0].lower() if isinstance(a[0], str) else a[0],
    "fprintf": lambda ip, a, n: [],
    "disp": lambda ip, a, n: [],
    "tic": lambda ip, a, n: [],
    "toc": lambda ip, a, n: mat(0.0),
    "drawnow": lambda ip, a, n: [],
    "size": _size,
    "numel": lambda ip, a, n: mat(float(len(a[0]) if isinstance(a[0], str) else mat(a[0]).size)),
    "length": lambda ip, a, n: mat(float(_length(a[0]))),
    "isempty": lambda ip, a, n: mat(len(a[0]) == 0 if isinstance(a[0], (str, list)) else False if isinstance(a[0], Struct) or is_callable(a[0])
                                    else mat(a[0]).size == 0),
    "zeros": lambda ip, a, n: np.zeros(_dims(a)) if a else mat(0.0),
    "ones": lambda ip, a, n: np.ones(_dims(a)) if a else mat(1.0),
    "true": lambda ip, a, n: np.ones(_dims(a), dtype=bool) if a else mat(True),
    "false": lambda ip, a, n: np.zeros(_dims(a), dtype=bool) if a else mat(False),
    "eye": lambda ip, a, n: np.eye(*_dims(a)),
    "logical": lambda ip, a, n: mat(a[0]) != 0 if mat(a[0]).dtype != bool else mat(a[0]),
    "double": lambda ip, a, n: num(mat(a[0])),
    "int32": lambda ip, a, n: np.clip(_round_half_away(num(mat(a[0]))), -2147483648.0, 2147483647.0),
    "isinf": lambda ip, a, n: np.isinf(num(mat(a[0]))),
    "imag": lambda ip, a, n: np.imag(num(mat(a[0]))).astype(np.float64),
    "real": lambda ip, a, n: np.real(num(mat(a[0]))).astype(np.float64),
    "isreal": lambda ip, a, n: mat(not np.iscomplexobj(mat(a[0]))),
    "struct": _struct,
    "roots": _roots,
    "linsolve": _linsolve,
    "polyval": lambda ip, a, n: np.polyval(num(mat(a[0])).reshape(-1), num(mat(a[1]))),
    "sum": _reduce(np.sum),
    "prod": _reduce(np.prod),
    "mean": _mean,
    "var": _var,
    "eig": _eig,
    "nthroot": lambda ip, a, n: _nthroot(num(mat(a[0])), num(mat(a[1]))),
    "cumsum": _cumsum,
    "exp": _elementwise(np.exp), "log": _elementwise(_log), "sqrt": _elementwise(_sqrt), "abs": _elementwise(np.abs),
    "floor": _elementwise(np.floor), "ceil": _elementwise(np.ceil), "round": _elementwise(_round_half_away),
    "isnan": lambda ip, a, n: np.isnan(num(mat(a[0]))),
    "power": lambda ip, a, n: ip.binop(".^", a[0], a[1]),
    "times": lambda ip, a, n: ip.binop(".*", a[0], a[1]),
    "mod": lambda ip, a, n: _mod(num(mat(a[0])), num(mat(a[1]))),
    "bsxfun": _bsxfun,
    "find": _find,
    "diag": _diag,
    "inv": lambda ip, a, n: np.linalg.inv(num(mat(a[0]))),
    "det": lambda ip, a, n: mat(np.linalg.det(num(mat(a[0])))),
    "chol": lambda ip, a, n: np.linalg.cholesky(num(mat(a[0]))).T,
    "svd": _svd,
    "reshape": _reshape,
    "repmat": _repmat,
    "squeeze": _squeeze,
    "norm": _norm,
    "max": _minmax(np.maximum, np.argmax),
    "min": _minmax(np.minimum, np.argmin),
    "sort": _sort,
    "eps": lambda ip, a, n: mat(np.spacing(np.abs(num(mat(a[0]))))) if a else mat(np.finfo(np.float64).eps),
    "pi": lambda ip, a, n: mat(math.pi),
    "inf": lambda ip, a, n: mat(math.inf),
    "Inf": lambda ip, a, n: mat(math.inf),
    "nan": lambda ip, a, n: mat(math.nan),
    "NaN": lambda ip, a, n: mat(math.nan),
    "strcmp": lambda ip, a, n: mat(isinstance(a[0], str) and isinstance(a[1], str) and a[0] == a[1]),
    "any": lambda ip, a, n: mat(bool(np.any(mat(a[0]) != 0))) if min(mat(a[0]).shape) == 1 else trim(np.any(mat(a[0]) != 0, axis=0, keepdims=True)),
    "all": lambda ip, a, n: mat(bool(np.all(mat(a[0]) != 0))) if min(mat(a[0]).shape) == 1 else trim(np.all(mat(a[0]) != 0, axis=0, keepdims=True)),
    "sparse": lambda ip, a, n: _sparse(a),
    "hist": _hist,
    "linspace": lambda ip, a, n: np.linspace(scalar(a[0]), scalar(a[1]), int(scalar(a[2])) if len(a) > 2 else 100).reshape(1, -1),
    "full": lambda ip, a, n: mat(a[0]),
}


def parse_args_builtin(args):
    """[a,b,...] = internal.stats.parseArgs(pnames, defaults, name1, value1, ...): the name / value pairs of predict.m:5-8"""
    pnames, defaults, rest = args[0], list(args[1]), args[2:]
    if len(rest) % 2:
        raise MError("parseArgs: name / value pairs expected")
    out = list(defaults)
    for q in range(0, len(rest), 2):
        key = rest[q]
        hits = [i for i, nm in enumerate(pnames) if nm.lower() == str(key).lower()]
        if not hits:                                                 # a unique, case-insensitive PREFIX of a name is accepted as well
            hits = [i for i, nm in enumerate(pnames) if nm.lower().startswith(str(key).lower())]   # (demo_2D.m:72 passes 'maxAttempt')
            if len(hits) > 1:
                raise MError("parseArgs: ambiguous parameter name %r" % (key,))
        if not hits:
            raise MError("parseArgs: unknown parameter %r" % (key,))
        out[hits[0]] = rest[q + 1]
    return out


# ------------------------------------------------------------------------------------------------ conveniences for the callers
def available(ref_dir=REF_DIR):
    return os.path.isdir(ref_dir) and os.path.exists(os.path.join(ref_dir, "GPz.m"))


def model_struct(m, d, k, method, heteroscedastic, g_dim, **extra):
    return Struct(m=mat(float(m)), d=mat(float(d)), k=mat(float(k)), method=method, heteroscedastic=mat(bool(heteroscedastic)),
                  g_dim=mat(float(g_dim)), **extra)


def col(x):
    return np.asarray(x, dtype=np.float64).reshape(-1, 1)


EMPTY = np.zeros((0, 0))
