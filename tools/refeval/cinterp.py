"""A small interpreter for the subset of C++ the reference's hot-path functions are written in.

BUILD-CONTAINER TOOL.  tools/make_ref_vectors.py uses it to EVALUATE functions of
/root/reference/src/*.cpp from their text (read at run time, never stored) on seeded inputs and to
write the resulting numbers to tests/golden/.  Nothing here is product code and nothing of the
reference's text is kept: only inputs and outputs (numbers) leave this tool.

Why an interpreter: the reference's translation units cannot be compiled in this image (every one of
them includes a MongoDB client header that is absent, and stand-ins are not allowed), but the functions
on the path -- modelWind, computeF, computeG, countG, dynamicConstraints, dynamicsGradients and the
mission cost / boundary functions -- are plain arithmetic over arrays and scalars.

Supported: declarations of scalars and fixed arrays (with initialisers), assignment and compound
assignment, ++/--, arithmetic with C semantics (integer division truncates, % follows the dividend,
int <-> double conversions on assignment and at calls), comparisons, && || !, ?:, casts, calls of
other interpreted member functions and of libm, member access, indexing, if / else, for, while,
switch / case / default / break, continue, return, try / catch, constructors (the body; base-class
constructors named in the initialiser list run first), `new T[n]`, vector resize / push_back / size / at.
Statements the path does not depend on (`auto t = std::chrono...;`, `cout << ...;`) are skipped.  A call of
something that is not in the interpreted sources and not libm (the MongoDB client the reference's constructor
tries to reach) raises ExternalUnavailable, which an enclosing try / catch of the interpreted code handles like
the exception the real library throws when there is no database -- so the reference's own catch block runs.
declare_members() reads a class declaration from a header and gives every data member its C++ default (NaN for
an uninitialised double).  A scalar declared without an initialiser
reads as NaN, which is how the reference's uninitialised `Gs` shows up as "undefined" in the output.
"""
import math
import re

NAN = float("nan")

TYPE_WORDS = {"int", "double", "float", "bool", "char", "long", "unsigned", "const", "FILE", "size_t", "short", "signed"}

_TOKEN = re.compile(r"""
    (?P<num>(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fFlLuU]*)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<chr>'(?:\\.|[^'\\])+')
  | (?P<op>\+\+|--|<<=|>>=|<<|>>|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|->|::|[-+*/%<>=!&|^~?:;,.(){}\[\]])
  | (?P<ws>\s+)
""", re.X)


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: " " * 0 + "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def tokenize(text):
    out, pos = [], 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise SyntaxError("cannot tokenize at %r" % text[pos:pos + 40])
        pos = m.end()
        kind = m.lastgroup
        if kind == "ws":
            continue
        out.append((kind, m.group(kind)))
    return out


class Break(Exception):
    pass


class Continue(Exception):
    pass


class Return(Exception):
    def __init__(self, value):
        self.value = value


class ExternalUnavailable(Exception):
    """The interpreted code called into a library that is not part of the interpreted sources."""


class Thrown(Exception):
    """A `throw` statement of the interpreted code."""


class Unavailable:
    """Value of a data member whose class is not interpreted (mongo::DBClientConnection): any use is an external call."""
    def __init__(self, what):
        self.what = what

    def __getattr__(self, name):
        raise ExternalUnavailable("%s.%s" % (self.__dict__.get("what", "?"), name))


class BigArray:
    """`new T[n]` / vector.resize(n) for an n too large for a Python list (the reference's neF x n work arrays at
    ts = 2000: 352 million entries, never touched on the path): elements are stored on first write."""
    def __init__(self, n, fill):
        self.n, self.fill, self.d = int(n), fill, {}

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self.d.get(k, self.fill) for k in range(*i.indices(self.n))]
        return self.d.get(i, self.fill)

    def __setitem__(self, i, v):
        self.d[i] = v


BIG = 40_000_000


def new_array(n, fill):
    return BigArray(n, fill) if n > BIG else [fill] * n


class Function:
    def __init__(self, cls, name, rettype, params, body):
        self.cls, self.name, self.rettype, self.params, self.body = cls, name, rettype, params, body


def _match_paren(text, i, open_ch, close_ch):
    depth = 1
    while depth and i < len(text):
        depth += {open_ch: 1, close_ch: -1}.get(text[i], 0)
        i += 1
    return i


def find_functions(text):
    """{(class, name): (rettype, params_text, body_text)} for every `T Class::name(...) {...}` at top level, and for
    every constructor `Class::Class(...) : inits {...}` its body under (class, class) plus the names its initialiser
    list calls under (class, "<bases>")."""
    text = strip_comments(text)
    found = {}
    pats = [(r"([A-Za-z_][\w:<>\*&\s]*?)\b(\w+)::(~?\w+)\s*\(", False), (r"()\b(\w+)::(\2)\s*\(", True)]
    for pat, ctor in pats:
        for m in re.finditer(pat, text):
            if (m.group(2) == m.group(3)) != ctor:
                continue
            start = m.start()
            if text.count("{", 0, start) != text.count("}", 0, start):      # must be at brace depth 0
                continue
            i = _match_paren(text, m.end(), "(", ")")
            params = text[m.end():i - 1]
            j = i
            while j < len(text) and text[j] not in "{;":
                j += 1
            if j >= len(text) or text[j] != "{":                          # a declaration
                continue
            inits = text[i:j]
            if ":" in inits and not ctor:
                continue
            k = _match_paren(text, j + 1, "{", "}")
            key = (m.group(2), m.group(3))
            if key in found:
                continue
            found[key] = ("void" if ctor else m.group(1).strip(), params, text[j:k])
            if ctor:
                found[(m.group(2), "<bases>")] = re.findall(r"[:,]\s*(\w+)\s*\(", inits)
    return found


def declare_members(header_text, cls):
    """{member: default} of the data members `class cls { ... }` declares in header_text: double -> its initialiser or
    NaN (uninitialised), int / bool -> initialiser or 0, std::string -> "", std::vector -> [], pointers -> [] (until a
    `new`), members of other class types -> Unavailable.  Member functions, nested types and access labels are skipped."""
    text = strip_comments(header_text)
    m = re.search(r"\bclass\s+" + re.escape(cls) + r"\b[^;{]*\{", text)
    if not m:
        raise NameError("class %s is not declared in this header" % cls)
    i, depth = m.end(), 1
    while depth and i < len(text):
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    body = text[m.end():i - 1]
    # drop nested struct / class bodies and inline function bodies
    out, depth = [], 0
    for ch in body:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                out.append(";")
            continue
        if depth == 0:
            out.append(ch)
    members = {}
    for stmt in "".join(out).split(";"):
        stmt = re.sub(r"\b(public|protected|private)\s*:", " ", stmt).strip()
        if not stmt or "(" in stmt.split("=")[0] or stmt.startswith(("struct", "class", "typedef", "using", "friend", "virtual")):
            continue
        stmt = re.sub(r"\b(const|static|mutable)\b", " ", stmt).strip()
        mt = re.match(r"((?:std::)?[A-Za-z_][\w:]*(?:\s*<.*>)?)\s+(.*)$", stmt, re.S)
        if not mt:
            continue
        ctype, decls = mt.group(1).strip(), mt.group(2)
        for d in decls.split(","):
            d = d.strip()
            if not d:
                continue
            init = None
            if "=" in d:
                d, init = (t.strip() for t in d.split("=", 1))
            ptr = d.startswith("*") or d.endswith("]")
            name = d.strip("*& ").split("[")[0].strip()
            if not re.fullmatch(r"[A-Za-z_]\w*", name):
                continue
            if ptr:
                val = []
            elif ctype in ("double", "float"):
                val = NAN if init is None else float({"M_PI": math.pi}.get(init, init))
            elif ctype in ("int", "bool", "long", "unsigned", "size_t", "short", "char"):
                val = 0 if init is None else int({"true": 1, "false": 0}.get(init, init))
            elif ctype in ("std::string", "string"):
                val = ""
            elif "vector" in ctype:
                val = []
            else:
                val = Unavailable(name)
            members[name] = val
    return members


class Parser:
    def __init__(self, tokens):
        self.t, self.i = tokens, 0

    # ---- token helpers
    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def accept(self, val):
        if self.peek()[1] == val and self.peek()[0] in ("op", "id"):
            self.i += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise SyntaxError("expected %r, found %r (token %d)" % (val, self.peek(), self.i))

    def skip_to_semicolon(self):
        depth = 0
        while True:
            k, v = self.next()
            if k == "eof":
                raise SyntaxError("unterminated statement")
            if v in "([{":
                depth += 1
            elif v in ")]}":
                depth -= 1
            elif v == ";" and depth == 0:
                return

    # ---- types
    def at_type(self):
        k, v = self.peek()
        if k != "id":
            return False
        if v in TYPE_WORDS or v == "auto":
            return True
        return v == "std" and self.peek(1)[1] == "::" and self.peek(2)[1] in ("string", "vector")

    def parse_type(self):
        words = []
        while True:
            k, v = self.peek()
            if k == "id" and v in TYPE_WORDS:
                words.append(v)
                self.i += 1
            elif v == "std" and self.peek(1)[1] == "::":
                self.i += 2
                words.append("std::" + self.next()[1])
                if self.accept("<"):
                    depth = 1
                    while depth:
                        v2 = self.next()[1]
                        depth += {"<": 1, ">": -1, ">>": -2}.get(v2, 0)
            else:
                break
        base = "double" if ("double" in words or "float" in words) else ("int" if any(w in words for w in ("int", "long", "short", "bool", "char", "unsigned", "size_t", "signed")) else (words[-1] if words else "int"))
        return base

    # ---- statements
    def parse_block_body(self):
        stmts = []
        while self.peek()[1] != "}" or self.peek()[0] != "op":
            if self.peek()[0] == "eof":
                raise SyntaxError("unterminated block")
            stmts.append(self.parse_statement())
        return stmts

    def parse_statement(self):
        k, v = self.peek()
        if k == "op" and v == "{":
            self.i += 1
            body = self.parse_block_body()
            self.expect("}")
            return ("block", body)
        if k == "op" and v == ";":
            self.i += 1
            return ("skip",)
        if k == "id":
            if v == "if":
                self.i += 1
                self.expect("(")
                c = self.parse_expr()
                self.expect(")")
                th = self.parse_statement()
                el = None
                if self.accept("else"):
                    el = self.parse_statement()
                return ("if", c, th, el)
            if v == "for":
                self.i += 1
                self.expect("(")
                init = ("skip",) if self.accept(";") else self.parse_simple_statement()
                cond = None if self.peek()[1] == ";" else self.parse_expr()
                self.expect(";")
                step = None if self.peek()[1] == ")" else self.parse_expr()
                self.expect(")")
                return ("for", init, cond, step, self.parse_statement())
            if v == "while":
                self.i += 1
                self.expect("(")
                c = self.parse_expr()
                self.expect(")")
                return ("while", c, self.parse_statement())
            if v == "switch":
                self.i += 1
                self.expect("(")
                e = self.parse_expr()
                self.expect(")")
                self.expect("{")
                arms = []
                while not self.accept("}"):
                    if self.accept("case"):
                        val = self.parse_expr()
                        self.expect(":")
                        arms.append((val, []))
                    elif self.accept("default"):
                        self.expect(":")
                        arms.append((None, []))
                    else:
                        arms[-1][1].append(self.parse_statement())
                return ("switch", e, arms)
            if v == "break":
                self.i += 1
                self.expect(";")
                return ("break",)
            if v == "continue":
                self.i += 1
                self.expect(";")
                return ("continue",)
            if v == "return":
                self.i += 1
                e = None if self.peek()[1] == ";" else self.parse_expr()
                self.expect(";")
                return ("return", e)
            if v == "throw":
                self.skip_to_semicolon()
                return ("throw",)
            if v == "try":
                self.i += 1
                self.expect("{")
                body = self.parse_block_body()
                self.expect("}")
                handlers = []
                while self.accept("catch"):
                    self.expect("(")
                    depth = 1
                    while depth:                      # the exception declaration: not needed
                        v2 = self.next()[1]
                        depth += {"(": 1, ")": -1}.get(v2, 0)
                    self.expect("{")
                    handlers.append(self.parse_block_body())
                    self.expect("}")
                return ("try", body, handlers)
            if v == "cout" or (v == "std" and self.peek(2)[1] in ("cout", "cerr")) or v in ("auto", "delete"):
                self.skip_to_semicolon()       # console output / chrono stopwatch: not on the path
                return ("skip",)
        return self.parse_simple_statement()

    def parse_simple_statement(self):
        """declaration or expression statement, including the terminating ';'"""
        if self.at_type():
            base = self.parse_type()
            decls = []
            while True:
                while self.accept("*") or self.accept("&"):
                    pass
                name = self.next()[1]
                dims = []
                while self.accept("["):
                    dims.append(None if self.peek()[1] == "]" else self.parse_expr())
                    self.expect("]")
                init = None
                if self.accept("="):
                    if self.accept("{"):
                        items = []
                        while not self.accept("}"):
                            items.append(self.parse_assign())
                            self.accept(",")
                        init = ("list", items)
                    else:
                        init = self.parse_assign()
                elif self.peek()[1] == "(" and self.peek()[0] == "op":      # vector<double> v(n) / v(n, value)
                    self.i += 1
                    args = []
                    while not self.accept(")"):
                        args.append(self.parse_assign())
                        self.accept(",")
                    init = ("ctor", args)
                decls.append((name, dims, init))
                if not self.accept(","):
                    break
            self.expect(";")
            return ("decl", base, decls)
        e = self.parse_expr()
        self.expect(";")
        return ("expr", e)

    # ---- expressions
    def parse_expr(self):
        e = self.parse_assign()
        while self.accept(","):
            e = ("comma", e, self.parse_assign())
        return e

    def parse_assign(self):
        lhs = self.parse_cond()
        k, v = self.peek()
        if k == "op" and v in ("=", "+=", "-=", "*=", "/=", "%="):
            self.i += 1
            return ("assign", v, lhs, self.parse_assign())
        return lhs

    def parse_cond(self):
        c = self.parse_binary(0)
        if self.accept("?"):
            a = self.parse_assign()
            self.expect(":")
            return ("cond", c, a, self.parse_assign())
        return c

    LEVELS = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"), ("*", "/", "%")]

    def parse_binary(self, lvl):
        if lvl == len(self.LEVELS):
            return self.parse_unary()
        e = self.parse_binary(lvl + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[lvl]:
            op = self.next()[1]
            e = ("binop", op, e, self.parse_binary(lvl + 1))
        return e

    def parse_unary(self):
        k, v = self.peek()
        if k == "op":
            if v in ("-", "+", "!", "~"):
                self.i += 1
                return ("unop", v, self.parse_unary())
            if v in ("++", "--"):
                self.i += 1
                return ("preinc", self.parse_unary(), 1 if v == "++" else -1)
            if v == "(" and self.peek(1)[0] == "id" and self.peek(1)[1] in TYPE_WORDS:
                save = self.i
                self.i += 1
                base = self.parse_type()
                if self.accept(")"):
                    return ("cast", base, self.parse_unary())
                self.i = save
        return self.parse_postfix()

    def parse_postfix(self):
        e = self.parse_primary()
        while True:
            k, v = self.peek()
            if k != "op":
                return e
            if v == "[":
                self.i += 1
                idx = self.parse_expr()
                self.expect("]")
                e = ("index", e, idx)
            elif v == "(":
                self.i += 1
                args = []
                while not self.accept(")"):
                    args.append(self.parse_assign())
                    self.accept(",")
                e = ("call", e, args)
            elif v in (".", "->"):
                self.i += 1
                e = ("member", e, self.next()[1])
            elif v in ("++", "--"):
                self.i += 1
                e = ("postinc", e, 1 if v == "++" else -1)
            else:
                return e

    def parse_primary(self):
        k, v = self.next()
        if k == "num":
            txt = v.rstrip("fFlLuU")
            if re.fullmatch(r"\d+", txt):
                return ("num", int(txt))
            return ("num", float(txt))
        if k == "str":
            return ("str", bytes(v[1:-1], "utf-8").decode("unicode_escape"))
        if k == "chr":
            return ("num", ord(bytes(v[1:-1], "utf-8").decode("unicode_escape")))
        if k == "id":
            if v == "std" and self.peek()[1] == "::":
                self.i += 1
                return ("name", self.next()[1])
            if v in ("true", "false"):
                return ("num", 1 if v == "true" else 0)
            if self.peek()[1] == "::" and self.peek()[0] == "op":       # a qualified name: lib::ns::function
                parts = [v]
                while self.accept("::"):
                    parts.append(self.next()[1])
                return ("name", "::".join(parts))
            if v == "new":                          # new T[count]
                base = self.parse_type()
                self.expect("[")
                cnt = self.parse_expr()
                self.expect("]")
                return ("newarr", base, cnt)
            return ("name", v)
        if k == "op" and v == "(":
            e = self.parse_expr()
            self.expect(")")
            return e
        raise SyntaxError("unexpected token %r" % ((k, v),))


def c_int(v):
    """C conversion of a value to int (truncation toward zero)."""
    if isinstance(v, float):
        return int(v) if math.isfinite(v) else 0
    return int(v)


def c_div(a, b):
    if isinstance(a, int) and isinstance(b, int):
        q = abs(a) // abs(b)
        return q if (a >= 0) == (b >= 0) else -q
    try:
        return a / b
    except ZeroDivisionError:
        a = float(a)
        if a != a or a == 0.0:
            return NAN
        neg = (a < 0) != (math.copysign(1.0, float(b)) < 0)
        return -math.inf if neg else math.inf


def c_mod(a, b):
    if isinstance(a, int) and isinstance(b, int):
        return a - b * c_div(a, b)
    return math.fmod(a, b)


def _pow(a, b):
    try:
        return math.pow(a, b)
    except (OverflowError, ValueError):
        return NAN


class Output:
    """What fopen/fprintf/fclose of the interpreted code produced: {filename: text}."""
    def __init__(self):
        self.files = {}


class Interp:
    def __init__(self, sources):
        """sources: list of C++ source texts (read by the caller from /root/reference at run time)."""
        self.raw = {}
        for text in sources:
            self.raw.update(find_functions(text))
        self.funcs = {}
        self.output = Output()
        self.calls = 0
        self.builtins = {
            "sin": math.sin, "cos": math.cos, "tan": math.tan, "sqrt": lambda v: math.sqrt(v) if v >= 0 else NAN,
            "atan2": math.atan2, "atan": math.atan, "asin": math.asin, "acos": math.acos, "exp": math.exp,
            "log": math.log, "pow": _pow, "fabs": abs, "abs": abs, "floor": lambda v: float(math.floor(v)),
            "ceil": lambda v: float(math.ceil(v)), "fmod": math.fmod, "max": max, "min": min,
            "fopen": self._fopen, "fprintf": self._fprintf, "fclose": lambda fp: 0,
        }
        self.constants = {"M_PI": math.pi, "NULL": 0}

    # ---- files written by the interpreted code
    def _fopen(self, name, mode):
        self.output.files[name] = ""
        return name

    def _fprintf(self, fp, fmt, *args):
        self.output.files[fp] += fmt % args
        return 0

    # ---- functions
    def function(self, classes, name):
        for cls in classes:
            key = (cls, name)
            if key in self.funcs:
                return self.funcs[key]
            if key in self.raw:
                ret, params, body = self.raw[key]
                plist = []
                for ptxt in [p.strip() for p in params.split(",") if p.strip()]:
                    ptoks = [t for t in tokenize(ptxt)]
                    is_arr = any(v in ("[", "*") for _, v in ptoks)
                    words = [v for k, v in ptoks if k == "id"]
                    pname = words[-1]
                    base = "double" if "double" in words[:-1] else "int"
                    plist.append((pname, base, is_arr))
                ps = Parser(tokenize(body))
                ps.expect("{")
                stmts = ps.parse_block_body()
                f = Function(cls, name, "void" if "void" in ret.split() else ("double" if "double" in ret else "int"), plist, stmts)
                self.funcs[key] = f
                return f
        return None

    def construct(self, obj, cls, args, between=None):
        """Run the constructor of class `cls` on `obj` the way C++ does: the base-class constructors its initialiser
        list names first (members' own constructors are the caller's business: their values are data), then its body.
        `between(obj)`, if given, runs after the base constructors and before the body (a test hook, e.g. to move the
        start position the reference hard-codes)."""
        for base in self.raw.get((cls, "<bases>"), []):
            if (base, base) in self.raw:
                self.construct(obj, base, args)
        if between is not None:
            between(obj)
        f = self.function([cls], cls)
        if f is None:
            raise NameError("no interpreted constructor %s::%s" % (cls, cls))
        self.calls += 1
        frame = Frame(self, obj, [{f.params[0][0]: args} if f.params else {}], [{}])
        try:
            frame.exec_block(f.body, new_scope=False)
        except Return:
            pass

    def call(self, obj, name, *args):
        """Call member function `name` on `obj` (obj._classes = most-derived first)."""
        f = self.function(obj._classes, name)
        if f is None:
            raise NameError("no interpreted function %s for %s" % (name, obj._classes))
        self.calls += 1
        scope = [{}]
        types = [{}]
        for (pname, base, is_arr), a in zip(f.params, args):
            if not is_arr:
                a = float(a) if base == "double" else c_int(a)
            scope[0][pname] = a
            types[0][pname] = None if is_arr else base
        frame = Frame(self, obj, scope, types)
        try:
            frame.exec_block(f.body, new_scope=False)
        except Return as r:
            if f.rettype == "double":
                return float(r.value) if r.value is not None else NAN
            if f.rettype == "int":
                return c_int(r.value)
            return None
        return None


class Frame:
    def __init__(self, it, obj, scopes, types):
        self.it, self.obj, self.scopes, self.types = it, obj, scopes, types

    # ---- name resolution: locals, then members of the object, then constants
    def lookup(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name]
        if hasattr(self.obj, name):
            return getattr(self.obj, name)
        if name in self.it.constants:
            return self.it.constants[name]
        if "::" in name:
            raise ExternalUnavailable(name)
        raise NameError(name)

    def store_name(self, name, value):
        for sc, ty in zip(reversed(self.scopes), reversed(self.types)):
            if name in sc:
                t = ty.get(name)
                sc[name] = float(value) if t == "double" else (c_int(value) if t == "int" else value)
                return sc[name]
        if hasattr(self.obj, name):
            cur = getattr(self.obj, name)
            if isinstance(cur, bool) or isinstance(cur, int):
                value = c_int(value)
            elif isinstance(cur, float):
                value = float(value)
            setattr(self.obj, name, value)
            return value
        raise NameError(name)

    # ---- statements
    def exec_block(self, stmts, new_scope=True):
        if new_scope:
            self.scopes.append({})
            self.types.append({})
        try:
            for s in stmts:
                self.exec(s)
        finally:
            if new_scope:
                self.scopes.pop()
                self.types.pop()

    def exec(self, s):
        kind = s[0]
        if kind == "expr":
            self.eval(s[1])
        elif kind == "decl":
            base = s[1]
            for name, dims, init in s[2]:
                if dims:
                    size = c_int(self.eval(dims[0])) if dims[0] is not None else len(init[1])
                    fill = NAN if base == "double" else 0
                    arr = [fill] * size
                    if init is not None and init[0] == "list":
                        vals = [self.eval(e) for e in init[1]]
                        for i2, v in enumerate(vals):
                            arr[i2] = float(v) if base == "double" else c_int(v)
                        for i2 in range(len(vals), size):     # C zero-fills the rest of an initialised array
                            arr[i2] = 0.0 if base == "double" else 0
                    self.scopes[-1][name] = arr
                    self.types[-1][name] = None
                elif base.startswith("std::vector"):
                    args = [self.eval(e) for e in init[1]] if init is not None and init[0] == "ctor" else []
                    self.scopes[-1][name] = [args[1] if len(args) > 1 else 0.0] * (c_int(args[0]) if args else 0)
                    self.types[-1][name] = None
                else:
                    if init is None:
                        v = NAN if base == "double" else 0
                    else:
                        v = self.eval(init)
                        if isinstance(v, (list, str)):      # pointer to an array / FILE*
                            base = "ptr"
                        elif base == "double":
                            v = float(v)
                        elif base == "int":
                            v = c_int(v)
                    self.scopes[-1][name] = v
                    self.types[-1][name] = base if base in ("int", "double") else None
        elif kind == "if":
            if self.truth(self.eval(s[1])):
                self.exec(s[2])
            elif s[3] is not None:
                self.exec(s[3])
        elif kind == "block":
            self.exec_block(s[1])
        elif kind == "for":
            self.scopes.append({})
            self.types.append({})
            try:
                self.exec(s[1])
                while s[2] is None or self.truth(self.eval(s[2])):
                    try:
                        self.exec(s[4])
                    except Break:
                        break
                    except Continue:
                        pass
                    if s[3] is not None:
                        self.eval(s[3])
            finally:
                self.scopes.pop()
                self.types.pop()
        elif kind == "while":
            while self.truth(self.eval(s[1])):
                try:
                    self.exec(s[2])
                except Break:
                    break
                except Continue:
                    pass
        elif kind == "switch":
            val = self.eval(s[1])
            arms = s[2]
            start = None
            for i2, (cv, _) in enumerate(arms):
                if cv is not None and self.eval(cv) == val:
                    start = i2
                    break
            if start is None:
                for i2, (cv, _) in enumerate(arms):
                    if cv is None:
                        start = i2
                        break
            if start is not None:
                self.scopes.append({})
                self.types.append({})
                try:
                    for _, body in arms[start:]:       # fall-through like C
                        for st in body:
                            self.exec(st)
                except Break:
                    pass
                finally:
                    self.scopes.pop()
                    self.types.pop()
        elif kind == "break":
            raise Break()
        elif kind == "continue":
            raise Continue()
        elif kind == "return":
            raise Return(None if s[1] is None else self.eval(s[1]))
        elif kind == "skip":
            pass
        elif kind == "throw":
            raise Thrown("the interpreted code threw an exception")
        elif kind == "try":
            try:
                self.exec_block(s[1])
            except (ExternalUnavailable, Thrown):
                # what the real library does when the service it needs is absent: throw; the first handler runs
                if not s[2]:
                    raise
                self.exec_block(s[2][0])
        else:
            raise RuntimeError("statement %r" % (kind,))

    @staticmethod
    def truth(v):
        return v != 0       # NaN compares unequal to 0, hence true, as in C

    # ---- lvalues
    def assign_to(self, lv, value):
        k = lv[0]
        if k == "name":
            return self.store_name(lv[1], value)
        if k == "index":
            base = self.eval(lv[1])
            idx = c_int(self.eval(lv[2]))
            if idx < 0 or idx >= len(base):
                raise IndexError("index %d outside an array of %d (interpreted code)" % (idx, len(base)))
            cur = base[idx]
            if isinstance(cur, float):
                value = float(value)
            elif isinstance(cur, int):
                value = c_int(value)
            base[idx] = value
            return value
        if k == "member":
            o = self.eval(lv[1])
            cur = getattr(o, lv[2])
            value = float(value) if isinstance(cur, float) else (c_int(value) if isinstance(cur, int) else value)
            setattr(o, lv[2], value)
            return value
        raise RuntimeError("not an lvalue: %r" % (lv,))

    # ---- expressions
    def eval(self, e):
        k = e[0]
        if k == "num" or k == "str":
            return e[1]
        if k == "name":
            return self.lookup(e[1])
        if k == "binop":
            op = e[1]
            if op == "&&":
                return 1 if (self.truth(self.eval(e[2])) and self.truth(self.eval(e[3]))) else 0
            if op == "||":
                return 1 if (self.truth(self.eval(e[2])) or self.truth(self.eval(e[3]))) else 0
            a, b = self.eval(e[2]), self.eval(e[3])
            if op == "+":
                return a + b
            if op == "-":
                return a - b
            if op == "*":
                return a * b
            if op == "/":
                return c_div(a, b)
            if op == "%":
                return c_mod(a, b)
            if op == "==":
                return 1 if a == b else 0
            if op == "!=":
                return 1 if a != b else 0
            if op == "<":
                return 1 if a < b else 0
            if op == ">":
                return 1 if a > b else 0
            if op == "<=":
                return 1 if a <= b else 0
            if op == ">=":
                return 1 if a >= b else 0
            raise RuntimeError("operator %s" % op)
        if k == "index":
            base = self.eval(e[1])
            idx = c_int(self.eval(e[2]))
            if idx < 0 or idx >= len(base):
                raise IndexError("index %d outside an array of %d (interpreted code)" % (idx, len(base)))
            return base[idx]
        if k == "member":
            return getattr(self.eval(e[1]), e[2])
        if k == "assign":
            op = e[1]
            if op == "=":
                return self.assign_to(e[2], self.eval(e[3]))
            cur, rhs = self.eval(e[2]), self.eval(e[3])
            if op == "+=":
                v = cur + rhs
            elif op == "-=":
                v = cur - rhs
            elif op == "*=":
                v = cur * rhs
            elif op == "/=":
                v = c_div(cur, rhs)
            else:
                v = c_mod(cur, rhs)
            return self.assign_to(e[2], v)
        if k == "unop":
            v = self.eval(e[2])
            if e[1] == "-":
                return -v
            if e[1] == "+":
                return v
            if e[1] == "!":
                return 0 if self.truth(v) else 1
            raise RuntimeError("unary %s" % e[1])
        if k == "call":
            fn = e[1]
            args = [self.eval(a) for a in e[2]]
            if fn[0] == "name":
                name = fn[1]
                if self.it.function(self.obj._classes, name) is not None:
                    return self.it.call(self.obj, name, *args)
                if name in self.it.builtins:
                    return self.it.builtins[name](*args)
                if "::" in name:
                    raise ExternalUnavailable(name)
                raise NameError("call of %s" % name)
            if fn[0] == "member":         # vector methods the path uses
                o = self.eval(fn[1])
                if fn[2] == "size":
                    return len(o)
                if fn[2] == "at":
                    return o[c_int(args[0])]
                if fn[2] == "push_back":
                    o.append(args[0])
                    return None
                if fn[2] == "resize":                  # std::vector<double>::resize(n): new elements are value-initialised (0.0)
                    n2 = c_int(args[0])
                    if n2 > BIG:
                        self.assign_to(fn[1], BigArray(n2, 0.0))
                    elif isinstance(o, list):
                        del o[n2:]
                        o.extend([0.0] * (n2 - len(o)))
                    else:
                        self.assign_to(fn[1], [0.0] * n2)
                    return None
            raise RuntimeError("call %r" % (fn,))
        if k == "postinc":
            cur = self.eval(e[1])
            self.assign_to(e[1], cur + e[2])
            return cur
        if k == "preinc":
            return self.assign_to(e[1], self.eval(e[1]) + e[2])
        if k == "cond":
            return self.eval(e[2]) if self.truth(self.eval(e[1])) else self.eval(e[3])
        if k == "cast":
            v = self.eval(e[2])
            return float(v) if e[1] == "double" else c_int(v)
        if k == "newarr":
            return new_array(c_int(self.eval(e[2])), NAN if e[1] == "double" else 0)
        if k == "comma":
            self.eval(e[1])
            return self.eval(e[2])
        raise RuntimeError("expression %r" % (k,))
