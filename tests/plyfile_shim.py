"""A minimal stand-in for the `plyfile` package (absent from this image; the reference imports it in scene/gaussian_model*.py and
scene/dataset_readers.py): exactly the API surface the reference uses --

    PlyElement.describe(structured_array, 'vertex'); PlyData([el]).write(path)
    PlyData.read(path); plydata.elements[0]['x']; plydata['vertex']; [p.name for p in element.properties]

-- with the file format plyfile itself produces for such calls on a little-endian host: a `format binary_little_endian 1.0`
header, one `element <name> <count>` line per element, one `property <type> <name>` line per field (type names char / uchar /
short / ushort / int / uint / float / double), `end_header`, then the records back to back.  TEST INFRASTRUCTURE: it lets the
reference's own save_ply / load_ply / storePly / fetchPly run, so that seganygaussians_amd/ply_io.py can be pinned against
files written and read by the reference's code paths."""
import numpy as np

_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}
_FROM_PLY = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
             "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
             "double": "f8", "float64": "f8"}


class PlyProperty:
    def __init__(self, name, val_dtype):
        self.name, self.val_dtype = name, val_dtype

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {self.val_dtype!r})"


class PlyElement:
    def __init__(self, name, properties, data):
        self.name, self.properties, self.data = name, properties, data

    @staticmethod
    def describe(data, name):
        if not isinstance(data, np.ndarray) or data.ndim != 1 or data.dtype.names is None:
            raise TypeError("only one-dimensional structured arrays are supported")
        props = []
        for n in data.dtype.names:
            t = data.dtype[n]
            if t.shape != () or t.kind not in "iuf":
                raise ValueError(f"unsupported field type for {n}: {t}")
            props.append(PlyProperty(n, t.kind + str(t.itemsize)))
        return PlyElement(name, props, data)

    @property
    def count(self):
        return len(self.data)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, key):
        return self.data[key]

    def header(self):
        lines = [f"element {self.name} {len(self.data)}"]
        lines += [f"property {_TO_PLY[p.val_dtype]} {p.name}" for p in self.properties]
        return lines


class PlyData:
    def __init__(self, elements=(), text=False, byte_order="=", comments=(), obj_info=()):
        if text:
            raise NotImplementedError("the reference only writes binary PLY files")
        self.elements = list(elements)
        self.comments = list(comments)

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def __contains__(self, name):
        return any(e.name == name for e in self.elements)

    def write(self, stream):
        lines = ["ply", "format binary_little_endian 1.0"] + [f"comment {c}" for c in self.comments]
        for e in self.elements:
            lines += e.header()
        lines.append("end_header")
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "wb") if own else stream
        try:
            f.write(("\n".join(lines) + "\n").encode("ascii"))
            for e in self.elements:
                le = np.dtype([(p.name, "<" + p.val_dtype) for p in e.properties])
                f.write(np.ascontiguousarray(e.data.astype(le, copy=False)).tobytes())
        finally:
            if own:
                f.close()

    @staticmethod
    def read(stream):
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "rb") if own else stream
        try:
            if f.readline().strip() != b"ply":
                raise ValueError("not a PLY file")
            fmt, specs = None, []
            while True:
                line = f.readline()
                if not line:
                    raise ValueError("PLY header without end_header")
                tok = line.decode("ascii").split()
                if not tok or tok[0] in ("comment", "obj_info"):
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] == "element":
                    specs.append((tok[1], int(tok[2]), []))
                elif tok[0] == "property":
                    if tok[1] == "list":
                        raise NotImplementedError("list properties")
                    specs[-1][2].append((tok[2], _FROM_PLY[tok[1]]))
                elif tok[0] == "end_header":
                    break
            elements = []
            for name, count, props in specs:
                if fmt == "ascii":
                    raw = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
                    data = np.empty(count, dtype=[(n, t) for n, t in props])
                    for i, (n, t) in enumerate(props):
                        data[n] = raw[:, i].astype(t)
                else:
                    order = "<" if fmt == "binary_little_endian" else ">"
                    dt = np.dtype([(n, order + t) for n, t in props])
                    data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count).copy()
                elements.append(PlyElement(name, [PlyProperty(n, t) for n, t in props], data))
            return PlyData(elements)
        finally:
            if own:
                f.close()
