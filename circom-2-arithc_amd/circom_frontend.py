"""A Circom-subset front-end: .circom text -> the calls the reference's unroller makes on `Compiler`.

Host-side producer of the flat gate list — NOT part of the accelerated path.  The reference's own
front-end is the iden3 parser (un-vendored Rust crates) plus src/process.rs / src/runtime.rs; neither can be built in
this image, so without this module the shipped input/circuit.circom (BASELINE config 0) and the reference's test
circuits could only be fed as hand-traced call sequences.  This is a small recursive-descent
parser for the language subset the reference supports (README.md:14-40) and a restatement, call for call, of what the
reference's unroller does with the resulting AST:

    program.rs:18-74     compile(): main component arguments, main body, IO discovery by name prefix
    process.rs:36-189    statements (declarations, substitutions, if / while, return, assert)
    process.rs:192-277   handle_substitution (variable / component / signal cases, array connections)
    process.rs:280-312   expressions; :315-419 calls; :426-478 infix -> gate; :485-533 prefix -> gate
    process.rs:536-579   get_signal_for_access / make_constant (variables become named constant signals)
    process.rs:649-764   execute_op (u32 folding), prefix -> infix (0 - x, 0 == x, u32::MAX ^ x)
    runtime.rs:56-420    context stack: clone-on-push inheritance, merge on pop (signals are NOT merged),
                         signal ids from one counter in declaration order, names "{ctx}.{name}[i]..."

Its output is the sequence of calls the unroller makes on `Compiler` — ("signal", id, name, value) /
("gate", AGateType, lhs id, rhs id, out id) / ("connect", a id, b id) — plus the input / output name prefixes of the
main template: exactly the "script" format of tests/golden/*.json and of the C++ CLI.  The iden3 parser's desugaring
is restated as documented in SURVEY Appendix A (`for` -> Block[init, While{cond, Block[body, step]}], `var x = e` ->
InitializationBlock[Declaration, Substitution], `i++` -> i = i + 1).  Random item names (`random_{u32}`,
runtime.rs:229) are replaced by a counter: they never reach circuit.txt / circuit_info.json and are filtered from
report.json (compiler.rs:519).
"""
from __future__ import annotations

import copy
import re
from typing import Dict, List, Optional, Tuple

U32_MAX = 0xFFFFFFFF
RETURN_VAR = "function_return_value"          # runtime.rs:16


class ProgramError(Exception):
    """Display strings of ProgramError / RuntimeError (program.rs:76-117, runtime.rs:795-817)."""


def runtime_error(msg: str) -> ProgramError:
    return ProgramError(f"Runtime error: {msg}")


# --------------------------------------------------------------------------------------------------------------------
# parser (subset of circom 2.x: pragma, templates, functions, signals / vars / components, for / while / if,
# substitutions, calls, array and component accesses, the infix / prefix operators of a_gate_type.rs:30-54)
# --------------------------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"""\s*(?:(//[^\n]*|/\*.*?\*/)|(\d+)|([A-Za-z_$][A-Za-z_0-9$]*)|(<==|==>|<--|-->|===|\*\*|<<|>>|<=|>=|==|!=|&&|\|\||\+\+|--|\+=|-=|\*=|[-+*/\\%<>=!~&|^(){}\[\];,.?:]))""", re.S)

INFIX = {"*": "Mul", "/": "Div", "+": "Add", "-": "Sub", "**": "Pow", "\\": "IntDiv", "%": "Mod", "<<": "ShiftL",
         ">>": "ShiftR", "<=": "LesserEq", ">=": "GreaterEq", "<": "Lesser", ">": "Greater", "==": "Eq", "!=": "NotEq",
         "||": "BoolOr", "&&": "BoolAnd", "|": "BitOr", "&": "BitAnd", "^": "BitXor"}
# a_gate_type.rs:30-54
GATE = {"Mul": "AMul", "Div": "ADiv", "Add": "AAdd", "Sub": "ASub", "Pow": "APow", "IntDiv": "AIntDiv", "Mod": "AMod",
        "ShiftL": "AShiftL", "ShiftR": "AShiftR", "LesserEq": "ALEq", "GreaterEq": "AGEq", "Lesser": "ALt",
        "Greater": "AGt", "Eq": "AEq", "NotEq": "ANeq", "BoolOr": "ABoolOr", "BoolAnd": "ABoolAnd", "BitOr": "ABitOr",
        "BitAnd": "ABitAnd", "BitXor": "AXor"}
# circom's grammar: comparison binds LOOSER than the bitwise operators
LEVELS = [["||"], ["&&"], ["==", "!=", "<", ">", "<=", ">="], ["|"], ["^"], ["&"], ["<<", ">>"], ["+", "-"],
          ["*", "/", "\\", "%"], ["**"]]


def tokenize(text: str) -> List[str]:
    out, pos = [], 0
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ProgramError("Parsing error")
        pos = m.end()
        if m.group(1) is None:
            out.append(m.group(2) or m.group(3) or m.group(4))
    return out


class Parser:
    def __init__(self, text: str):
        self.t = tokenize(text)
        self.i = 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else None

    def eat(self, tok=None):
        cur = self.peek()
        if cur is None or (tok is not None and cur != tok):
            raise ProgramError("Parsing error")
        self.i += 1
        return cur

    # ---- program
    def program(self):
        templates, functions, main = {}, {}, None
        while self.peek() is not None:
            if self.peek() == "pragma":
                while self.eat() != ";":
                    pass
            elif self.peek() in ("template", "function"):
                kind = self.eat()
                name = self.eat()
                self.eat("(")
                params = []
                while self.peek() != ")":
                    params.append(self.eat())
                    if self.peek() == ",":
                        self.eat()
                self.eat(")")
                body = self.block()
                (templates if kind == "template" else functions)[name] = {"params": params, "body": body["stmts"]}
            elif self.peek() == "component" and self.peek(1) == "main":
                self.eat(); self.eat()
                if self.peek() == "{":                    # {public [...]}: no meaning for this compiler
                    while self.eat() != "}":
                        pass
                self.eat("=")
                main = self.expr()
                self.eat(";")
            else:
                raise ProgramError("Parsing error")
        for t in templates.values():                       # template_data.get_inputs() / get_outputs(), declaration order
            t["inputs"], t["outputs"] = [], []
            self._collect_io(t["body"], t)
        return templates, functions, main

    def _collect_io(self, stmts, t):
        for s in stmts:
            if s["k"] == "decl" and s["xtype"] == "signal" and s["io"]:
                t["inputs" if s["io"] == "input" else "outputs"].append(s["name"])
            for key in ("stmts", "inits"):
                if key in s:
                    self._collect_io(s[key], t)
            for key in ("then", "else", "body"):
                if s.get(key):
                    self._collect_io([s[key]], t)

    # ---- statements
    def block(self):
        self.eat("{")
        stmts = []
        while self.peek() != "}":
            stmts.append(self.statement())
        self.eat("}")
        return {"k": "block", "stmts": stmts}

    def statement(self):
        p = self.peek()
        if p == "{":
            return self.block()
        if p in ("signal", "var", "component"):
            s = self.declaration()
            self.eat(";")
            return s
        if p == "for":
            self.eat(); self.eat("(")
            init = self.declaration() if self.peek() in ("var",) else self.simple()
            self.eat(";")
            cond = self.expr()
            self.eat(";")
            step = self.simple()
            self.eat(")")
            body = self.statement()
            # the iden3 parser's desugaring: Block[init, While{cond, Block[body, step]}]
            return {"k": "block", "stmts": [init, {"k": "while", "cond": cond, "body": {"k": "block", "stmts": [body, step]}}]}
        if p == "while":
            self.eat(); self.eat("(")
            cond = self.expr()
            self.eat(")")
            return {"k": "while", "cond": cond, "body": self.statement()}
        if p == "if":
            self.eat(); self.eat("(")
            cond = self.expr()
            self.eat(")")
            then = self.statement()
            els = None
            if self.peek() == "else":
                self.eat()
                els = self.statement()
            return {"k": "if", "cond": cond, "then": then, "else": els}
        if p == "return":
            self.eat()
            v = self.expr()
            self.eat(";")
            return {"k": "return", "value": v}
        if p == "assert":
            self.eat(); self.eat("(")
            v = self.expr()
            self.eat(")"); self.eat(";")
            return {"k": "assert", "arg": v}
        s = self.simple()
        self.eat(";")
        return s

    def declaration(self):
        xtype = self.eat()
        io = None
        if xtype == "signal" and self.peek() in ("input", "output"):
            io = self.eat()
        inits = []
        while True:
            name = self.eat()
            dims = []
            while self.peek() == "[":
                self.eat()
                dims.append(self.expr())
                self.eat("]")
            inits.append({"k": "decl", "xtype": xtype, "io": io, "name": name, "dims": dims})
            if self.peek() in ("=", "<=="):
                op = self.eat()
                rhe = self.expr()
                inits.append({"k": "subst", "var": name, "access": [], "op": "AssignVar" if op == "=" else "AssignConstraintSignal", "rhe": rhe})
            if self.peek() == ",":
                self.eat()
                continue
            break
        return {"k": "init", "inits": inits}

    def simple(self):
        """substitution-like statements: x = e, x <== e, e ==> x, x++, x += e, a === b"""
        lhs = self.expr()
        op = self.peek()
        if op in ("=", "<==", "<--"):
            self.eat()
            rhe = self.expr()
            return self._subst(lhs, {"=": "AssignVar", "<==": "AssignConstraintSignal", "<--": "AssignSignal"}[op], rhe)
        if op in ("==>", "-->"):
            self.eat()
            target = self.expr()
            return self._subst(target, "AssignConstraintSignal" if op == "==>" else "AssignSignal", lhs)
        if op in ("++", "--"):
            self.eat()
            return self._subst(lhs, "AssignVar", {"k": "infix", "op": "Add" if op == "++" else "Sub", "l": lhs, "r": {"k": "num", "v": 1}})
        if op in ("+=", "-=", "*="):
            self.eat()
            rhe = self.expr()
            return self._subst(lhs, "AssignVar", {"k": "infix", "op": INFIX[op[0]], "l": lhs, "r": rhe})
        if op == "===":
            self.eat()
            self.expr()
            return {"k": "unsupported"}
        raise ProgramError("Parsing error")

    @staticmethod
    def _subst(target, op, rhe):
        if target["k"] != "var":
            raise ProgramError("Parsing error")
        return {"k": "subst", "var": target["name"], "access": target["access"], "op": op, "rhe": rhe}

    # ---- expressions
    def expr(self, level=0):
        if level == len(LEVELS):
            return self.prefix()
        lhs = self.expr(level + 1)
        while self.peek() in LEVELS[level]:
            op = self.eat()
            rhs = self.expr(level + 1)
            lhs = {"k": "infix", "op": INFIX[op], "l": lhs, "r": rhs}
        return lhs

    def prefix(self):
        if self.peek() in ("-", "!", "~"):
            op = self.eat()
            return {"k": "prefix", "op": {"-": "Sub", "!": "BoolNot", "~": "Complement"}[op], "r": self.prefix()}
        return self.atom()

    def atom(self):
        p = self.peek()
        if p == "(":
            self.eat()
            e = self.expr()
            self.eat(")")
            return e
        if p is not None and p.isdigit():
            return {"k": "num", "v": int(self.eat())}
        name = self.eat()
        if not re.match(r"[A-Za-z_$]", name):
            raise ProgramError("Parsing error")
        if self.peek() == "(":
            self.eat()
            args = []
            while self.peek() != ")":
                args.append(self.expr())
                if self.peek() == ",":
                    self.eat()
            self.eat(")")
            return {"k": "call", "id": name, "args": args}
        access = []
        while self.peek() in ("[", "."):
            if self.eat() == "[":
                access.append(("array", self.expr()))
                self.eat("]")
            else:
                access.append(("component", self.eat()))
        return {"k": "var", "name": name, "access": access}


# --------------------------------------------------------------------------------------------------------------------
# runtime.rs: contexts
# --------------------------------------------------------------------------------------------------------------------
def nested_get(value, path):                       # runtime.rs get_nested_value
    cur = value
    for idx in path:
        if not isinstance(cur, list):
            raise runtime_error("Access Error")
        if idx >= len(cur):
            raise runtime_error("Index out of bounds")
        cur = cur[idx]
    return cur


class Context:
    def __init__(self, name: str):
        self.ctx_name = name
        self.names = set()
        self.variables: Dict[str, object] = {}
        self.signals: Dict[str, object] = {}
        self.components: Dict[str, object] = {}

    def inherit(self) -> "Context":                # Context::new_with_inheritance (a deep clone)
        c = Context(self.ctx_name)
        c.names = set(self.names)
        c.variables = copy.deepcopy(self.variables)
        c.signals = copy.deepcopy(self.signals)
        c.components = copy.deepcopy(self.components)
        return c

    def merge(self, child: "Context"):             # Context::merge: signals are NOT merged
        for name, var in child.variables.items():
            if name in self.variables:
                self.variables[name] = copy.deepcopy(var)
        if RETURN_VAR in child.variables:
            self.variables[RETURN_VAR] = copy.deepcopy(child.variables[RETURN_VAR])
        for name, comp in child.components.items():
            if name in self.components:
                self.components[name] = copy.deepcopy(comp)

    def data_type(self, name: str) -> str:
        if name in self.variables:
            return "Variable"
        if name in self.signals:
            return "Signal"
        if name in self.components:
            return "Component"
        raise runtime_error(f"Item not declared: get_item_data_type: {name}")


class Unroller:
    """program.rs::compile + process.rs, recording the calls made on `Compiler`."""

    def __init__(self, text: str):
        self.templates, self.functions, self.main = Parser(text).program()
        self.contexts: List[Context] = [Context("0")]          # Runtime::new
        self.next_signal_id = 0
        self.random = 0
        self.script: List[list] = []

    # ---- runtime helpers
    @property
    def ctx(self) -> Context:
        return self.contexts[0]

    def push(self, inherit: bool, name: str):
        self.contexts.insert(0, self.ctx.inherit() if inherit else Context(name))

    def pop(self, merge: bool):
        child = self.contexts.pop(0)
        if merge and self.contexts:
            self.ctx.merge(child)

    def gen_signal(self) -> int:
        sid = self.next_signal_id
        self.next_signal_id += 1
        return sid

    def declare(self, data_type: str, name: str, dims: List[int]):
        ctx = self.ctx
        if name in ctx.names and data_type != "Variable":
            raise runtime_error("Item already declared")
        ctx.names.add(name)

        def nest(leaf, ds):
            return leaf() if not ds else [nest(leaf, ds[1:]) for _ in range(ds[0])]
        if data_type == "Signal":
            ctx.signals[name] = nest(self.gen_signal, dims)
        elif data_type == "Variable":
            ctx.variables[name] = nest(lambda: None, dims)
        else:
            ctx.components[name] = nest(dict, dims)

    def declare_random(self, data_type: str):
        name = f"random_{self.random}"
        self.random += 1
        self.declare(data_type, name, [])
        return (name, [])

    @staticmethod
    def access_str(ctx_name: str, access) -> str:          # DataAccess::access_str
        name, path = access
        s = f"{ctx_name}.{name}"
        for kind, v in path:
            s += f"[{v}]" if kind == "array" else f".{v}"
        return s

    @staticmethod
    def idx(path) -> List[int]:                             # access_to_u32
        out = []
        for kind, v in path:
            if kind != "array":
                raise runtime_error("Access Error")
            out.append(v)
        return out

    def var_value(self, access) -> Optional[int]:
        name, path = access
        if name not in self.ctx.variables:
            raise runtime_error(f"Item not declared: get_variable_value: {access}")
        v = nested_get(self.ctx.variables[name], self.idx(path))
        if isinstance(v, list):
            raise runtime_error("Data Item content is not a single value")
        return v

    def need_value(self, access) -> int:
        v = self.var_value(access)
        if v is None:
            raise ProgramError("Empty data item")
        return v

    def set_variable(self, access, value):
        name, path = access
        if name not in self.ctx.variables:
            raise runtime_error(f"Item not declared: set_variable: {access}")
        p = self.idx(path)
        if not p:
            if isinstance(self.ctx.variables[name], list):
                raise runtime_error("Data Item content is not a single value")
            self.ctx.variables[name] = value
            return
        holder = nested_get(self.ctx.variables[name], p[:-1])
        if not isinstance(holder, list):
            raise runtime_error("Access Error")
        if p[-1] >= len(holder):
            raise runtime_error("Index out of bounds")
        if isinstance(holder[p[-1]], list):
            raise runtime_error("Data Item content is not a single value")
        holder[p[-1]] = value

    def signal_content(self, access):
        name, path = access
        if name not in self.ctx.signals:
            raise runtime_error(f"Item not declared: get_signal_content: {access}")
        return nested_get(self.ctx.signals[name], self.idx(path))

    def signal_id(self, access) -> int:
        v = self.signal_content(access)
        if isinstance(v, list):
            raise runtime_error("Data Item content is not a single value")
        return v

    @staticmethod
    def split_component_access(access):                     # process_component_access
        name, path = access
        initial, final, signal = [], [], None
        for kind, v in path:
            if kind == "array":
                (final if signal is not None else initial).append(v)
            else:
                if signal is not None:
                    raise runtime_error("Access Error")
                signal = v
        if signal is None:
            raise runtime_error("Access Error")
        return (name, initial), (signal, final)

    def component_signal_content(self, access):
        (cname, cpath), (sname, spath) = self.split_component_access(access)
        if cname not in self.ctx.components:
            raise runtime_error(f"Item not declared: get_component_signal_id: {access}")
        m = nested_get(self.ctx.components[cname], cpath)
        if isinstance(m, list):
            raise runtime_error("Data Item content is not a single value")
        if sname not in m:
            raise runtime_error(f"Item not declared: get_signal_id: {sname}")
        return nested_get(m[sname], spath)

    def component_signal_id(self, access) -> int:
        v = self.component_signal_content(access)
        if isinstance(v, list):
            raise runtime_error("Data Item content is not a single value")
        return v

    # ---- Compiler calls
    def add_signal(self, sid: int, name: str, value: Optional[int]):
        self.script.append(["signal", sid, name, value])

    def add_gate(self, op: str, a: int, b: int, o: int):
        self.script.append(["gate", GATE[op], a, b, o])

    def add_connection(self, a: int, b: int):
        self.script.append(["connect", a, b])

    # ---- program.rs::compile
    def compile(self):
        if self.main is None or self.main["k"] != "call":
            raise ProgramError("Main expression not a call")
        name = self.main["id"]
        t = self.templates[name]
        values = [self.var_value(self.expression(e)) for e in self.main["args"]]
        for pname, v in zip(t["params"], values):
            self.declare("Variable", pname, [])
            self.set_variable((pname, []), v)
        self.statements(t["body"])
        return {"script": self.script, "input_prefixes": list(t["inputs"]), "output_prefixes": list(t["outputs"])}

    # ---- process.rs
    def statements(self, stmts):
        for s in stmts:
            self.statement(s)

    def statement(self, s):
        k = s["k"]
        if k == "init":
            self.statements(s["inits"])
        elif k == "block":
            self.statements(s["stmts"])
        elif k == "subst":
            self.substitution(s)
        elif k == "decl":
            data_type = {"signal": "Signal", "var": "Variable", "component": "Component"}[s["xtype"]]
            dim_access = [self.expression(e) for e in s["dims"]]
            dims = [self.need_value(a) for a in dim_access]
            self.declare(data_type, s["name"], dims)
            if data_type == "Signal":
                ctx = self.ctx
                if not dims:
                    self.add_signal(self.signal_id((s["name"], [])), self.access_str(ctx.ctx_name, (s["name"], [])), None)
                else:
                    indices = [0] * len(dims)
                    while True:
                        acc = (s["name"], [("array", i) for i in indices])
                        self.add_signal(self.signal_id(acc), self.access_str(ctx.ctx_name, acc), None)
                        carry = True                     # increment_indices
                        for d in range(len(dims) - 1, -1, -1):
                            if carry:
                                if indices[d] < dims[d] - 1:
                                    indices[d] += 1
                                    carry = False
                                else:
                                    indices[d] = 0
                        if carry:
                            break
        elif k == "if":
            result = self.need_value(self.expression(s["cond"]))
            if result == 0:
                if s["else"] is not None:
                    self.push(True, "IF_FALSE")
                    self.statement(s["else"])
                    self.pop(True)
            else:
                self.push(True, "IF_TRUE")
                self.statement(s["then"])
                self.pop(True)
        elif k == "while":
            self.push(True, "WHILE_PRE")
            while True:
                if self.need_value(self.expression(s["cond"])) == 0:
                    break
                self.push(True, "WHILE_EXE")
                self.statement(s["body"])
                self.pop(True)
            self.pop(True)
        elif k == "return":
            value = self.need_value(self.expression(s["value"]))
            self.declare("Variable", RETURN_VAR, [])
            self.set_variable((RETURN_VAR, []), value)
        elif k == "assert":
            if self.need_value(self.expression(s["arg"])) == 0:
                raise runtime_error("Assertion failed")
        else:
            raise ProgramError("Statement not implemented")

    def build_access(self, name, access):
        path = []
        for kind, v in access:
            if kind == "array":
                path.append(("array", self.need_value(self.expression(v))))
            else:
                path.append(("component", v))
        return (name, path)

    def signal_content_for_access(self, access):
        dt = self.ctx.data_type(access[0])
        if dt == "Signal":
            return self.signal_content(access)
        if dt == "Component":
            return self.component_signal_content(access)
        raise ProgramError("Invalid data type")

    def connect_arrays(self, a, b):
        if len(a) != len(b):
            raise ProgramError("Invalid data type")
        for x, y in zip(a, b):
            if isinstance(x, list) != isinstance(y, list):
                raise ProgramError("Invalid data type")
            if isinstance(x, list):
                self.connect_arrays(x, y)
            else:
                self.add_connection(x, y)

    def substitution(self, s):                               # process.rs:192-277
        lh = self.build_access(s["var"], s["access"])
        rh = self.expression(s["rhe"])
        dt = self.ctx.data_type(s["var"])
        if dt == "Variable":
            self.set_variable(lh, self.var_value(rh))
        elif dt == "Component":
            if s["op"] == "AssignVar":
                name, path = rh
                m = nested_get(self.ctx.components[name], self.idx(path))
                if isinstance(m, list):
                    raise runtime_error("Data Item content is not a single value")
                cname, cpath = lh
                p = self.idx(cpath)
                if not p:
                    if isinstance(self.ctx.components[cname], list):
                        raise runtime_error("Data Item content is not a single value")
                    self.ctx.components[cname] = copy.deepcopy(m)
                else:
                    holder = nested_get(self.ctx.components[cname], p[:-1])
                    if not isinstance(holder, list):
                        raise runtime_error("Access Error")
                    if p[-1] >= len(holder):
                        raise runtime_error("Index out of bounds")
                    holder[p[-1]] = copy.deepcopy(m)
            elif s["op"] == "AssignConstraintSignal":
                content = self.component_signal_content(lh)
                if isinstance(content, list):
                    assigned = self.signal_content_for_access(rh)
                    if not isinstance(assigned, list):
                        raise ProgramError("Invalid data type")
                    self.connect_arrays(content, assigned)
                else:
                    component_signal = self.component_signal_id(lh)
                    assigned = self.signal_for_access(rh)
                    self.add_connection(assigned, component_signal)
            else:
                raise ProgramError("Operation not supported")
        else:
            kind = s["rhe"]["k"]
            if kind == "var":
                content = self.signal_content(lh)
                if isinstance(content, list):
                    assigned = self.signal_content_for_access(rh)
                    if not isinstance(assigned, list):
                        raise ProgramError("Invalid data type")
                    self.connect_arrays(content, assigned)
                else:
                    self.add_connection(self.signal_for_access(rh), content)
            elif kind in ("call", "infix", "prefix", "num"):
                given = self.signal_id(lh)
                self.add_connection(self.signal_for_access(rh), given)
            else:
                raise ProgramError("Signal substitution not implemented")

    def expression(self, e):                                 # process.rs:280-312
        k = e["k"]
        if k == "call":
            return self.call(e["id"], e["args"])
        if k == "infix":
            return self.infix(e["op"], e["l"], e["r"])
        if k == "prefix":
            return self.prefix_op(e["op"], e["r"])
        if k == "num":
            if e["v"] > U32_MAX:
                raise ProgramError("Parsing error")
            acc = self.declare_random("Variable")
            self.set_variable(acc, e["v"])
            return acc
        if k == "var":
            return self.build_access(e["name"], e["access"])
        raise ProgramError("Expression not implemented")

    def call(self, name, args):                              # process.rs:315-419
        is_function = name in self.functions
        if is_function:
            data = self.functions[name]
        elif name in self.templates:
            data = self.templates[name]
        else:
            raise ProgramError("Undefined function or template")
        values = [self.need_value(self.expression(a)) for a in args]
        self.push(False, name)
        for pname, v in zip(data["params"], values):
            self.declare("Variable", pname, [])
            self.set_variable((pname, []), v)
        self.statements(data["body"])
        function_return, component_return = None, {}
        if is_function:
            if RETURN_VAR in self.ctx.variables:
                function_return = self.ctx.variables[RETURN_VAR]
        else:
            for sname in data["inputs"] + data["outputs"]:
                if sname not in self.ctx.signals:
                    raise runtime_error(f"Item not declared: get_signal: {sname}")
                component_return[sname] = copy.deepcopy(self.ctx.signals[sname])
        self.pop(False)
        ret = (f"{name}_{RETURN_VAR}_{self.random}", [])
        self.random += 1
        if is_function:
            self.declare("Variable", ret[0], [])
            self.set_variable(ret, function_return)
        else:
            self.declare("Component", ret[0], [])
            self.ctx.components[ret[0]] = component_return
        return ret

    def infix(self, op, lhe, rhe):                           # process.rs:426-478
        la = self.expression(lhe)
        ra = self.expression(rhe)
        lt, rt = self.ctx.data_type(la[0]), self.ctx.data_type(ra[0])
        if lt == "Variable" and rt == "Variable":
            res = execute_op(self.need_value(la), self.need_value(ra), op)
            acc = self.declare_random("Variable")
            self.set_variable(acc, res)
            return acc
        lhs_id = self.signal_for_access(la)
        rhs_id = self.signal_for_access(ra)
        out = self.declare_random("Signal")
        out_id = self.signal_id(out)
        self.add_signal(out_id, self.access_str(self.ctx.ctx_name, out), None)
        self.add_gate(op, lhs_id, rhs_id, out_id)
        return out

    def prefix_op(self, op, rhe):                            # process.rs:485-533, :758-764
        ra = self.expression(rhe)
        lhs_value, infix_op = {"Sub": (0, "Sub"), "BoolNot": (0, "Eq"), "Complement": (U32_MAX, "BitXor")}[op]
        if self.ctx.data_type(ra[0]) == "Variable":
            res = execute_op(lhs_value, self.need_value(ra), infix_op)
            acc = self.declare_random("Variable")
            self.set_variable(acc, res)
            return acc
        lhs_id = self.make_constant(lhs_value)
        rhs_id = self.signal_for_access(ra)
        out = self.declare_random("Signal")
        out_id = self.signal_id(out)
        self.add_signal(out_id, self.access_str(self.ctx.ctx_name, out), None)
        self.add_gate(infix_op, lhs_id, rhs_id, out_id)
        return out

    def signal_for_access(self, access) -> int:              # process.rs:536-556
        dt = self.ctx.data_type(access[0])
        if dt == "Signal":
            return self.signal_id(access)
        if dt == "Variable":
            return self.make_constant(self.need_value(access))
        return self.component_signal_id(access)

    def make_constant(self, value: int) -> int:              # process.rs:558-579
        name = f"const_signal_{value}"
        if name in self.ctx.signals and not isinstance(self.ctx.signals[name], list):
            return self.ctx.signals[name]
        self.declare("Signal", name, [])
        sid = self.signal_id((name, []))
        self.add_signal(sid, self.access_str(self.ctx.ctx_name, (name, [])), value)
        return sid


def execute_op(lhs: int, rhs: int, op: str) -> int:          # process.rs:649-750 (u32; Rust release-mode wrap-around)
    if op == "Mul":
        return (lhs * rhs) & U32_MAX
    if op in ("Div", "IntDiv"):
        if rhs == 0:
            raise ProgramError("Operation error: " + ("Division by zero" if op == "Div" else "Integer division by zero"))
        return lhs // rhs
    if op == "Add":
        return (lhs + rhs) & U32_MAX
    if op == "Sub":
        if lhs < rhs:
            raise ProgramError("Operation error: Subtraction underflow")
        return lhs - rhs
    if op == "Pow":
        return pow(lhs, rhs, 1 << 32)
    if op == "Mod":
        if rhs == 0:
            raise ProgramError("Operation error: Modulo by zero")
        return lhs % rhs
    if op == "ShiftL":
        return (lhs << (rhs & 31)) & U32_MAX
    if op == "ShiftR":
        return lhs >> (rhs & 31)
    table = {"LesserEq": lhs <= rhs, "GreaterEq": lhs >= rhs, "Lesser": lhs < rhs, "Greater": lhs > rhs, "Eq": lhs == rhs,
             "NotEq": lhs != rhs, "BoolOr": lhs != 0 or rhs != 0, "BoolAnd": lhs != 0 and rhs != 0}
    if op in table:
        return 1 if table[op] else 0
    return {"BitOr": lhs | rhs, "BitAnd": lhs & rhs, "BitXor": lhs ^ rhs}[op]


def unroll(text: str) -> dict:
    """.circom text -> {"script", "input_prefixes", "output_prefixes"} (raises ProgramError like compile())."""
    return Unroller(text).compile()
