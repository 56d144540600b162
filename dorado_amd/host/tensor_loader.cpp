// dorado_amd/host/tensor_loader.cpp — see tensor_loader.h.
#include "tensor_loader.h"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>

namespace dorado_amd::host {

namespace {

[[noreturn]] void die(const std::string &path, const std::string &why) {
    throw std::runtime_error("load_tensor_file(" + path + "): " + why);
}

uint16_t rd16(const uint8_t *p) { return uint16_t(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t *p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }
uint64_t rd64(const uint8_t *p) { return uint64_t(rd32(p)) | uint64_t(rd32(p + 4)) << 32; }

struct ZipEntry {
    size_t offset = 0, size = 0;   // of the stored payload
    bool stored = true;            // method 0 (TorchScript compresses only its code/ entries)
};

// Central directory walk (PKWARE APPNOTE 4.3.12, 4.3.16; zip64 extra field 0x0001 when sizes are 0xffffffff).
std::map<std::string, ZipEntry> zip_entries(const std::vector<uint8_t> &f, const std::string &path) {
    if (f.size() < 22) die(path, "too small for a zip archive");
    size_t eocd = std::string::npos;
    for (size_t i = f.size() - 22;; --i) {
        if (rd32(&f[i]) == 0x06054b50u) {
            eocd = i;
            break;
        }
        if (i == 0 || f.size() - i > 66000) break;
    }
    if (eocd == std::string::npos) die(path, "not a zip archive (no end-of-central-directory record)");
    if (eocd + 22 > f.size()) die(path, "truncated end-of-central-directory record");
    uint64_t n = rd16(&f[eocd + 10]), cd_off = rd32(&f[eocd + 16]);
    if (cd_off == 0xffffffffu || n == 0xffffu) {   // zip64
        if (eocd < 20 || rd32(&f[eocd - 20]) != 0x07064b50u) die(path, "zip64 locator missing");
        const uint64_t e64 = rd64(&f[eocd - 20 + 8]);
        if (e64 > f.size() || f.size() - e64 < 56 || rd32(&f[e64]) != 0x06064b50u) die(path, "bad zip64 end record");
        n = rd64(&f[e64 + 32]);
        cd_off = rd64(&f[e64 + 48]);
    }
    std::map<std::string, ZipEntry> out;
    if (cd_off > f.size() || n > f.size() / 46) die(path, "corrupt central directory");
    size_t p = cd_off;
    for (uint64_t i = 0; i < n; ++i) {
        if (p > f.size() || f.size() - p < 46 || rd32(&f[p]) != 0x02014b50u) die(path, "corrupt central directory");
        const uint16_t method = rd16(&f[p + 10]);
        uint64_t csize = rd32(&f[p + 20]), usize = rd32(&f[p + 24]), lho = rd32(&f[p + 42]);
        const uint16_t nlen = rd16(&f[p + 28]), xlen = rd16(&f[p + 30]), clen = rd16(&f[p + 32]);
        if (f.size() - p - 46 < size_t(nlen) + xlen + clen) die(path, "corrupt central directory (names run past the end)");
        const std::string name(reinterpret_cast<const char *>(&f[p + 46]), nlen);
        size_t x = p + 46 + nlen;
        const size_t xend = x + xlen;
        while (x + 4 <= xend) {
            const uint16_t id = rd16(&f[x]), sz = rd16(&f[x + 2]);
            if (x + 4 + sz > xend) die(path, "corrupt extra field of '" + name + "'");
            if (id == 0x0001) {
                size_t q = x + 4;
                const size_t qend = x + 4 + sz;
                auto take64 = [&](uint64_t &v) {
                    if (q + 8 > qend) die(path, "short zip64 extra field of '" + name + "'");
                    v = rd64(&f[q]);
                    q += 8;
                };
                if (usize == 0xffffffffu) take64(usize);
                if (csize == 0xffffffffu) take64(csize);
                if (lho == 0xffffffffu) take64(lho);
            }
            x += 4 + sz;
        }
        if (lho > f.size() || f.size() - lho < 30 || rd32(&f[lho]) != 0x04034b50u)
            die(path, "corrupt local header of '" + name + "'");
        const size_t data = lho + 30 + rd16(&f[lho + 26]) + rd16(&f[lho + 28]);
        const uint64_t stored_size = method == 0 ? usize : csize;
        if (data > f.size() || stored_size > f.size() - data) die(path, "entry '" + name + "' runs past the end of the file");
        out[name] = {data, size_t(method == 0 ? usize : csize), method == 0};
        p = xend + clen;
    }
    return out;
}

// ---- the pickle subset TorchScript's pickler emits ----
struct Obj;
using P = std::shared_ptr<Obj>;
struct Obj {
    enum Kind { NONE, BOOL, INT, STR, GLOBAL, TUPLE, LIST, DICT, OBJECT, STORAGE, TENSOR } kind = NONE;
    int64_t i = 0;
    std::string s;                                   // STR, GLOBAL ("module name")
    std::vector<P> items;                            // TUPLE / LIST
    std::vector<std::pair<P, P>> dict;               // DICT / OBJECT state
    // STORAGE: s = dtype class name, key in items[0]->s, numel in i.  TENSOR: items = {storage}, shape/stride/offset:
    std::vector<int64_t> shape, stride;
    int64_t offset = 0;
};
P mk(Obj::Kind k) {
    auto o = std::make_shared<Obj>();
    o->kind = k;
    return o;
}

struct Unpickler {
    const uint8_t *p, *end;
    const std::string &path;
    std::vector<P> stack;
    std::vector<size_t> marks;
    std::map<uint32_t, P> memo;

    P pop() {
        if (stack.empty()) die(path, "pickle stack underflow");
        P v = stack.back();
        stack.pop_back();
        return v;
    }
    const P &top() {
        if (stack.empty()) die(path, "pickle stack underflow");
        return stack.back();
    }
    std::vector<P> pop_mark() {
        if (marks.empty()) die(path, "pickle MARK missing");
        const size_t m = marks.back();
        marks.pop_back();
        if (m > stack.size()) die(path, "pickle MARK beyond the stack");
        std::vector<P> v(stack.begin() + long(m), stack.end());
        stack.resize(m);
        return v;
    }
    void need(size_t n) {
        if (size_t(end - p) < n) die(path, "truncated pickle");
    }
    std::string line() {
        const uint8_t *q = p;
        while (q < end && *q != '\n') ++q;
        if (q == end) die(path, "truncated pickle");
        std::string s(reinterpret_cast<const char *>(p), size_t(q - p));
        p = q + 1;
        return s;
    }
    static std::vector<int64_t> ints(const P &t, const std::string &path) {
        if (t->kind != Obj::TUPLE) die(path, "expected a tuple of ints");
        std::vector<int64_t> v;
        for (auto &e : t->items) {
            if (e->kind != Obj::INT) die(path, "expected a tuple of ints");
            v.push_back(e->i);
        }
        return v;
    }
    P reduce(const P &fn, const P &args) {
        if (fn->kind != Obj::GLOBAL || args->kind != Obj::TUPLE) die(path, "unsupported REDUCE");
        if (fn->s == "torch._utils _rebuild_tensor_v2" || fn->s == "torch._utils _rebuild_tensor") {
            if (args->items.size() < 4 || args->items[0]->kind != Obj::STORAGE) die(path, "malformed _rebuild_tensor_v2");
            P t = mk(Obj::TENSOR);
            t->items = {args->items[0]};
            if (args->items[1]->kind != Obj::INT) die(path, "malformed storage offset");
            t->offset = args->items[1]->i;
            t->shape = ints(args->items[2], path);
            t->stride = ints(args->items[3], path);
            return t;
        }
        if (fn->s == "collections OrderedDict") return mk(Obj::DICT);
        if (fn->s == "torch._utils _rebuild_parameter") {   // Parameter(data, requires_grad, backward_hooks)
            if (args->items.empty() || args->items[0]->kind != Obj::TENSOR) die(path, "malformed _rebuild_parameter");
            return args->items[0];
        }
        die(path, "unsupported callable in pickle: " + fn->s);
    }
    P run() {
        while (true) {
            need(1);
            const uint8_t op = *p++;
            switch (op) {
                case 0x80: need(1); ++p; break;                                   // PROTO
                case '.': return pop();                                            // STOP
                case '(': marks.push_back(stack.size()); break;                    // MARK
                case ')': stack.push_back(mk(Obj::TUPLE)); break;                  // EMPTY_TUPLE
                case '}': stack.push_back(mk(Obj::DICT)); break;                   // EMPTY_DICT
                case ']': stack.push_back(mk(Obj::LIST)); break;                   // EMPTY_LIST
                case 'N': stack.push_back(mk(Obj::NONE)); break;                   // NONE
                case 0x88: case 0x89: { P b = mk(Obj::BOOL); b->i = (op == 0x88); stack.push_back(b); break; }
                case 'K': { need(1); P v = mk(Obj::INT); v->i = *p++; stack.push_back(v); break; }            // BININT1
                case 'M': { need(2); P v = mk(Obj::INT); v->i = rd16(p); p += 2; stack.push_back(v); break; }  // BININT2
                case 'J': { need(4); P v = mk(Obj::INT); v->i = int32_t(rd32(p)); p += 4; stack.push_back(v); break; }
                case 0x8a: {                                                       // LONG1
                    need(1);
                    const int n = *p++;
                    need(size_t(n));
                    if (n > 8) die(path, "LONG1 wider than 64 bits");
                    uint64_t u = 0;
                    for (int k = 0; k < n; ++k) u |= uint64_t(p[k]) << (8 * k);
                    if (n > 0 && n < 8 && (p[n - 1] & 0x80)) u |= ~uint64_t(0) << (8 * n);
                    p += n;
                    P v = mk(Obj::INT);
                    v->i = int64_t(u);
                    stack.push_back(v);
                    break;
                }
                case 'X': {                                                        // BINUNICODE
                    need(4);
                    const uint32_t n = rd32(p);
                    p += 4;
                    need(n);
                    P v = mk(Obj::STR);
                    v->s.assign(reinterpret_cast<const char *>(p), n);
                    p += n;
                    stack.push_back(v);
                    break;
                }
                case 'c': {                                                        // GLOBAL
                    P g = mk(Obj::GLOBAL);
                    const std::string mod = line(), name = line();
                    g->s = mod + " " + name;
                    stack.push_back(g);
                    break;
                }
                case 'q': need(1); memo[*p++] = top(); break;                      // BINPUT
                case 'r': need(4); memo[rd32(p)] = top(); p += 4; break;           // LONG_BINPUT
                case 'h': { need(1); auto it = memo.find(*p++); if (it == memo.end()) die(path, "bad BINGET"); stack.push_back(it->second); break; }
                case 'j': { need(4); auto it = memo.find(rd32(p)); p += 4; if (it == memo.end()) die(path, "bad LONG_BINGET"); stack.push_back(it->second); break; }
                case 't': { P t = mk(Obj::TUPLE); t->items = pop_mark(); stack.push_back(t); break; }          // TUPLE
                case 0x85: case 0x86: case 0x87: {                                 // TUPLE1..3
                    const size_t n = size_t(op - 0x84);
                    if (stack.size() < n) die(path, "pickle stack underflow");
                    P t = mk(Obj::TUPLE);
                    t->items.assign(stack.end() - long(n), stack.end());
                    stack.resize(stack.size() - n);
                    stack.push_back(t);
                    break;
                }
                case 'e': { auto v = pop_mark(); if (stack.empty() || stack.back()->kind != Obj::LIST) die(path, "APPENDS on non-list"); for (auto &x : v) stack.back()->items.push_back(x); break; }
                case 'a': { P v = pop(); if (stack.empty() || stack.back()->kind != Obj::LIST) die(path, "APPEND on non-list"); stack.back()->items.push_back(v); break; }
                case 'u': {                                                        // SETITEMS
                    auto v = pop_mark();
                    if (stack.empty() || (stack.back()->kind != Obj::DICT && stack.back()->kind != Obj::OBJECT) || (v.size() & 1)) die(path, "malformed SETITEMS");
                    for (size_t k = 0; k < v.size(); k += 2) stack.back()->dict.emplace_back(v[k], v[k + 1]);
                    break;
                }
                case 's': {                                                        // SETITEM
                    P val = pop(), key = pop();
                    if (stack.empty() || (stack.back()->kind != Obj::DICT && stack.back()->kind != Obj::OBJECT)) die(path, "malformed SETITEM");
                    stack.back()->dict.emplace_back(key, val);
                    break;
                }
                case 0x81: { pop(); pop(); stack.push_back(mk(Obj::OBJECT)); break; }                          // NEWOBJ(cls, args)
                case 'b': {                                                        // BUILD(obj, state)
                    P state = pop();
                    if (stack.empty()) die(path, "pickle stack underflow");
                    if (state->kind == Obj::DICT) for (auto &kv : state->dict) stack.back()->dict.push_back(kv);
                    break;
                }
                case 'R': { P args = pop(), fn = pop(); stack.push_back(reduce(fn, args)); break; }            // REDUCE
                case 'Q': {                                                        // BINPERSID: ('storage', StorageClass, key, device, numel)
                    P pid = pop();
                    if (pid->kind != Obj::TUPLE || pid->items.size() < 5 || pid->items[0]->kind != Obj::STR || pid->items[0]->s != "storage" ||
                        pid->items[1]->kind != Obj::GLOBAL || pid->items[2]->kind != Obj::STR || pid->items[4]->kind != Obj::INT)
                        die(path, "unsupported persistent id");
                    P st = mk(Obj::STORAGE);
                    st->s = pid->items[1]->s;
                    st->items = {pid->items[2]};
                    st->i = pid->items[4]->i;
                    stack.push_back(st);
                    break;
                }
                default: {
                    char buf[8];
                    snprintf(buf, sizeof buf, "0x%02x", op);
                    die(path, std::string("unsupported pickle opcode ") + buf);
                }
            }
        }
    }
};

struct DInfo {
    DType t;
    size_t size;
};
DInfo storage_dtype(const std::string &cls, const std::string &path) {
    static const std::map<std::string, DInfo> m = {
            {"torch HalfStorage", {DType::F16, 2}},   {"torch BFloat16Storage", {DType::BF16, 2}},
            {"torch FloatStorage", {DType::F32, 4}},  {"torch DoubleStorage", {DType::F64, 8}},
            {"torch CharStorage", {DType::I8, 1}},    {"torch ByteStorage", {DType::U8, 1}},
            {"torch ShortStorage", {DType::I16, 2}},  {"torch IntStorage", {DType::I32, 4}},
            {"torch LongStorage", {DType::I64, 8}},   {"torch BoolStorage", {DType::BOOL, 1}},
    };
    const auto it = m.find(cls);
    if (it == m.end()) die(path, "unsupported storage class " + cls);
    return it->second;
}

float half_to_float(uint16_t h) {
    const uint32_t sign = uint32_t(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, bits;
    if (e == 0) {
        if (m == 0) {
            bits = sign;
        } else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            bits = sign | uint32_t(127 - 15 - sh + 1) << 23 | (m & 0x3ffu) << 13;
        }
    } else if (e == 31) {
        bits = sign | 0x7f800000u | m << 13;
    } else {
        bits = sign | (e + 127 - 15) << 23 | m << 13;
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

}  // namespace

size_t LoadedTensor::numel() const {
    size_t n = 1;
    for (int64_t d : shape) n *= size_t(d);
    return n;
}

std::vector<float> LoadedTensor::to_float() const {
    const size_t n = numel();
    std::vector<float> out(n);
    const uint8_t *p = data.data();
    for (size_t i = 0; i < n; ++i) {
        switch (dtype) {
            case DType::F16: { uint16_t h; std::memcpy(&h, p + 2 * i, 2); out[i] = half_to_float(h); break; }
            case DType::BF16: { uint16_t h; std::memcpy(&h, p + 2 * i, 2); const uint32_t b = uint32_t(h) << 16; std::memcpy(&out[i], &b, 4); break; }
            case DType::F32: std::memcpy(&out[i], p + 4 * i, 4); break;
            case DType::F64: { double d; std::memcpy(&d, p + 8 * i, 8); out[i] = float(d); break; }
            case DType::I8: out[i] = float(int8_t(p[i])); break;
            case DType::U8: case DType::BOOL: out[i] = float(p[i]); break;
            case DType::I16: { int16_t v; std::memcpy(&v, p + 2 * i, 2); out[i] = float(v); break; }
            case DType::I32: { int32_t v; std::memcpy(&v, p + 4 * i, 4); out[i] = float(v); break; }
            case DType::I64: { int64_t v; std::memcpy(&v, p + 8 * i, 8); out[i] = float(v); break; }
        }
    }
    return out;
}

std::vector<LoadedTensor> load_tensor_file(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) die(path, "cannot open");
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const auto entries = zip_entries(f, path);
    std::string root;
    for (const auto &kv : entries) {
        const std::string &n = kv.first;
        if (n.size() > 9 && n.compare(n.size() - 9, 9, "/data.pkl") == 0) root = n.substr(0, n.size() - 8);
    }
    if (root.empty()) die(path, "no <name>/data.pkl entry (not a TorchScript archive)");
    const ZipEntry pk = entries.at(root + "data.pkl");
    if (!pk.stored) die(path, "data.pkl is compressed; TorchScript archives store tensor data uncompressed");
    Unpickler u{&f[pk.offset], &f[pk.offset] + pk.size, path, {}, {}, {}};
    const P top = u.run();
    std::vector<std::pair<P, P>> attrs;
    if (top->kind == Obj::OBJECT || top->kind == Obj::DICT) {
        attrs = top->dict;
    } else if (top->kind == Obj::LIST || top->kind == Obj::TUPLE) {   // torch.save([tensors])-like payloads
        for (size_t i = 0; i < top->items.size(); ++i) {
            P k = mk(Obj::STR);
            k->s = std::to_string(i);
            attrs.emplace_back(k, top->items[i]);
        }
    } else if (top->kind == Obj::TENSOR) {
        P k = mk(Obj::STR);
        k->s = "0";
        attrs.emplace_back(k, top);
    } else {
        die(path, "unexpected pickle payload");
    }
    std::vector<LoadedTensor> out;
    for (const auto &kv : attrs) {
        if (kv.second->kind != Obj::TENSOR) continue;   // e.g. "training": False
        const Obj &t = *kv.second;
        if (t.items.empty() || t.items[0]->kind != Obj::STORAGE || t.items[0]->items.empty()) die(path, "tensor without storage");
        const Obj &st = *t.items[0];
        const DInfo di = storage_dtype(st.s, path);
        const auto it = entries.find(root + "data/" + st.items[0]->s);
        if (it == entries.end()) die(path, "storage '" + st.items[0]->s + "' missing from the archive");
        if (!it->second.stored) die(path, "storage '" + st.items[0]->s + "' is compressed; TorchScript archives store tensor data uncompressed");
        LoadedTensor lt;
        lt.name = kv.first->kind == Obj::STR ? kv.first->s : std::to_string(out.size());
        lt.dtype = di.t;
        lt.shape = t.shape;
        if (t.stride.size() != t.shape.size()) die(path, "shape / stride rank mismatch");
        const size_t es = di.size;
        // bounds, overflow-checked: rank, element count (a stride-0 view may not blow the output up beyond the
        // archive's own size class), largest linear index touched
        if (t.shape.size() > 16) die(path, "tensor rank above 16");
        const uint64_t storage_elems = it->second.size / es;
        uint64_t n64 = 1, maxidx = 0;
        if (t.offset < 0) die(path, "negative storage offset");
        maxidx = uint64_t(t.offset);
        for (size_t d = 0; d < t.shape.size(); ++d) {
            if (t.shape[d] < 0 || t.stride[d] < 0) die(path, "negative shape / stride");
            const uint64_t sd = uint64_t(t.shape[d]), st_d = uint64_t(t.stride[d]);
            if (sd != 0 && n64 > (uint64_t(1) << 40) / sd) die(path, "tensor '" + lt.name + "' is implausibly large");
            n64 *= sd;
            if (sd > 1) {
                if (st_d != 0 && (sd - 1) > (uint64_t(1) << 62) / st_d) die(path, "tensor '" + lt.name + "' reaches outside its storage");
                maxidx += (sd - 1) * st_d;
                if (maxidx >= (uint64_t(1) << 62)) die(path, "tensor '" + lt.name + "' reaches outside its storage");
            }
        }
        const size_t n = size_t(n64);
        if (n > 0 && maxidx >= storage_elems) die(path, "tensor '" + lt.name + "' reaches outside its storage");
        if (n > 0 && n64 > 64 * std::max<uint64_t>(storage_elems, 1)) die(path, "tensor '" + lt.name + "' expands its storage more than 64x");
        lt.data.resize(n * es);
        const uint8_t *src = &f[it->second.offset];
        // contiguous fast path
        bool contig = true;
        int64_t expect = 1;
        for (size_t d = t.shape.size(); d-- > 0;) {
            if (t.shape[d] != 1 && t.stride[d] != expect) contig = false;
            expect *= t.shape[d];
        }
        if (contig) {
            if (n) std::memcpy(lt.data.data(), src + size_t(t.offset) * es, n * es);
        } else {
            std::vector<int64_t> idx(t.shape.size(), 0);
            for (size_t k = 0; k < n; ++k) {
                int64_t lin = t.offset;
                for (size_t d = 0; d < idx.size(); ++d) lin += idx[d] * t.stride[d];
                std::memcpy(lt.data.data() + k * es, src + size_t(lin) * es, es);
                for (size_t d = idx.size(); d-- > 0;) {
                    if (++idx[d] < t.shape[d]) break;
                    idx[d] = 0;
                }
            }
        }
        out.push_back(std::move(lt));
    }
    if (out.empty()) die(path, "archive holds no tensor");
    return out;
}

std::vector<std::string> lstm_model_tensor_names(int n_convs, int lstm_layers, bool flstm, bool linear_bias,
                                                 bool decomposition) {
    // basecall/crf_utils.cpp:26-88
    const std::vector<std::string> conv_names{".conv.weight.tensor", ".conv.bias.tensor"};
    const std::vector<std::string> lstm_names =
            flstm ? std::vector<std::string>{".rnn.dn_weight_ih.tensor", ".rnn.dn_weight_hh.tensor",
                                             ".rnn.up_weight_ih.tensor", ".rnn.up_weight_hh.tensor",
                                             ".rnn.up_bias_ih.tensor",   ".rnn.up_bias_hh.tensor"}
                  : std::vector<std::string>{".rnn.weight_ih_l0.tensor", ".rnn.weight_hh_l0.tensor",
                                             ".rnn.bias_ih_l0.tensor", ".rnn.bias_hh_l0.tensor"};
    std::vector<std::string> t;
    for (int cv = 0; cv < n_convs; ++cv)
        for (const auto &n : conv_names) t.push_back(std::to_string(cv) + n);
    for (int l = 0; l < lstm_layers; ++l)
        for (const auto &n : lstm_names) t.push_back(std::to_string(n_convs + l + 1) + n);   // skip the fused layer index
    const int layer = n_convs + lstm_layers + 1;
    t.push_back(std::to_string(layer) + ".linear.weight.tensor");
    if (linear_bias) t.push_back(std::to_string(layer) + ".linear.bias.tensor");
    if (decomposition) t.push_back(std::to_string(layer + 1) + ".linear.weight.tensor");
    return t;
}

std::vector<std::string> tx_model_tensor_names(int n_convs, int depth) {
    // basecall/crf_utils.cpp:90-150
    std::vector<std::string> t;
    for (int cv = 0; cv < n_convs; ++cv)
        for (const char *n : {".conv.weight.tensor", ".conv.bias.tensor"}) t.push_back("conv." + std::to_string(cv) + n);
    for (int e = 0; e < depth; ++e)
        for (const char *n : {".self_attn.Wqkv.weight.tensor", ".self_attn.out_proj.weight.tensor",
                              ".self_attn.out_proj.bias.tensor", ".ff.fc1.weight.tensor", ".ff.fc2.weight.tensor",
                              ".norm1.weight.tensor", ".norm2.weight.tensor"})
            t.push_back("transformer_encoder." + std::to_string(e) + n);
    for (const char *n : {"upsample.linear.weight.tensor", "upsample.linear.bias.tensor", "crf.linear.weight.tensor"})
        t.push_back(n);
    return t;
}

}  // namespace dorado_amd::host

// ---- C entry points for the Python tests / loaders ----
using namespace dorado_amd::host;
static thread_local std::string g_terr;
static thread_local std::vector<LoadedTensor> g_loaded;

extern "C" {
const char *mibch_tensor_last_error(void) { return g_terr.c_str(); }
// Loads `path`; returns the number of tensors (kept in a thread-local until the next call) or -1.
int mibch_tensor_open(const char *path) {
    try {
        g_loaded = load_tensor_file(path);
        return int(g_loaded.size());
    } catch (const std::exception &e) {
        g_terr = e.what();
        g_loaded.clear();
        return -1;
    }
}
// dtype code (DType order), rank, shape (up to 8 dims), numel; name copied into name_out (<= 63 chars).
int mibch_tensor_info(int idx, int *dtype, int *rank, int64_t *shape8, int64_t *numel, char *name_out) {
    if (idx < 0 || size_t(idx) >= g_loaded.size()) return -1;
    const LoadedTensor &t = g_loaded[size_t(idx)];
    *dtype = int(t.dtype);
    *rank = int(t.shape.size());
    for (size_t d = 0; d < t.shape.size() && d < 8; ++d) shape8[d] = t.shape[d];
    *numel = int64_t(t.numel());
    std::snprintf(name_out, 64, "%s", t.name.c_str());
    return 0;
}
int mibch_tensor_copy_raw(int idx, void *dst, uint64_t bytes) {
    if (idx < 0 || size_t(idx) >= g_loaded.size() || bytes != g_loaded[size_t(idx)].data.size()) return -1;
    std::memcpy(dst, g_loaded[size_t(idx)].data.data(), bytes);
    return 0;
}
int mibch_tensor_copy_float(int idx, float *dst, uint64_t numel) {
    if (idx < 0 || size_t(idx) >= g_loaded.size() || numel != g_loaded[size_t(idx)].numel()) return -1;
    const auto v = g_loaded[size_t(idx)].to_float();
    std::memcpy(dst, v.data(), v.size() * 4);
    return 0;
}
// '\n'-separated file names in parameter order; returns the byte length needed (incl. NUL).
int mibch_model_tensor_names(int is_tx, int n_convs, int layers, int flstm, int linear_bias, int decomposition,
                             char *out, int cap) {
    const auto v = is_tx ? tx_model_tensor_names(n_convs, layers)
                         : lstm_model_tensor_names(n_convs, layers, flstm != 0, linear_bias != 0, decomposition != 0);
    std::string s;
    for (const auto &n : v) s += n + "\n";
    if (int(s.size()) + 1 <= cap) std::memcpy(out, s.c_str(), s.size() + 1);
    return int(s.size()) + 1;
}
}
