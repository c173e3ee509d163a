// vdl_parse.cpp -- text front door: VDL lines -> Program.
// One statement per line, "<id>,<Op>,<fields...>", operands print as "Id <n>"
// (/root/reference/src/Vdl.hs:97-99,410-453,476-477).  Everything from ";;" on is the
// --metadata suffix (Vdl.hs:463-466) that eval_query.sh:20 strips; it is ignored here.
#include "vdl_ir.h"
#include "vdl.h"

#include <cctype>
#include <cstring>

namespace vdl {

const char *const kBinNames[B_COUNT] = {"LogicalAnd", "LogicalOr", "BitwiseAnd", "BitwiseOr", "BitShift", "Equals",
                                        "Add", "Subtract", "Greater", "Multiply", "Divide", "Modulo"};

const char *op_name(Op op, int bin) {
    switch (op) {
    case Op::Load: return "Load";
    case Op::Project: return "Project";
    case Op::RangeV: return "RangeV";
    case Op::RangeC: return "RangeC";
    case Op::Binary: return (bin >= 0 && bin < B_COUNT) ? kBinNames[bin] : "Binary";
    case Op::FoldSelect: return "FoldSelect";
    case Op::FoldSum: return "FoldSum";
    case Op::FoldMin: return "FoldMin";
    case Op::FoldMax: return "FoldMax";
    case Op::FoldChoose: return "FoldChoose";
    case Op::FoldCount: return "FoldCount";
    case Op::Gather: return "Gather";
    case Op::Scatter: return "Scatter";
    case Op::Partition: return "Partition";
    case Op::Shuffle: return "Shuffle";
    case Op::Materialize: return "MaterializeCompact";
    case Op::Like: return "Like";
    case Op::Cross: return bin ? "CrossProductInner" : "CrossProductOuter";
    case Op::Semisort: return "Semisort";
    }
    return "?";
}

namespace {

[[noreturn]] void bad(int code, int line, const std::string &msg) {
    throw Error(code, "line " + std::to_string(line) + ": " + msg);
}

std::string trim(const std::string &s) {
    size_t b = 0, e = s.size();
    while (b < e && isspace((unsigned char)s[b])) b++;
    while (e > b && isspace((unsigned char)s[e - 1])) e--;
    return s.substr(b, e - b);
}

std::vector<std::string> split(const std::string &s) {
    std::vector<std::string> out;
    size_t p = 0;
    for (;;) {
        size_t q = s.find(',', p);
        if (q == std::string::npos) { out.push_back(trim(s.substr(p))); break; }
        out.push_back(trim(s.substr(p, q - p)));
        p = q + 1;
    }
    return out;
}

int64_t to_int(const std::string &s, int line) {
    if (s.empty()) bad(VDL_ERR_PARSE, line, "empty integer field");
    errno = 0;
    char *end = nullptr;
    long long v = strtoll(s.c_str(), &end, 10);
    if (*end || errno) bad(VDL_ERR_PARSE, line, "bad integer '" + s + "'");
    return (int64_t)v;
}

int to_ref(const std::string &s, int line) {
    if (s.compare(0, 3, "Id ") != 0) bad(VDL_ERR_PARSE, line, "expected 'Id <n>', got '" + s + "'");
    int64_t v = to_int(trim(s.substr(3)), line);
    if (v <= 0 || v > (1 << 24)) bad(VDL_ERR_PARSE, line, "operand id out of range");
    return (int)v;
}

}  // namespace

// The VLite dialect ("lighter syntax (one value per vector)", /root/reference/src/MainFuns.hs:70), printed by
// toVList / printLine (Vdl.hs:370-408,455-475): operands in the same order as the VDL lines, no field names,
//   <id>,Load,<name> | <id>,Project,Id v | <id>,RangeV,<from>,Id v,<step> | <id>,RangeC,<from>,<count>,<step>
//   <id>,<BinOp|Fold*|Partition|Gather>,Id a,Id b | <id>,Scatter,Id src,Id fold,Id pos | <id>,Semisort,Id v
//   <id>,Shuffle,Id v | <id>,Like,Id data,Id dict,<pattern> | <id>,CrossProduct{Outer,Inner},Id l,Id r
//   <id>,Output,Id v   or   <name>,Output,<display type>,Id v   (a named output does not print its id: previous + 1)
static Program parse_vlite_program(const char *text, size_t len) {
    Program P;
    P.nodes.resize(64);
    int lineno = 0, last_id = 0;
    size_t pos = 0;
    auto use = [&](int id, int line) {
        if (id <= 0 || (size_t)id >= P.nodes.size() || P.nodes[(size_t)id].id == 0)
            bad(VDL_ERR_PARSE, line, "reference to undefined vector Id " + std::to_string(id));
    };
    while (pos <= len) {
        size_t eol = pos;
        while (eol < len && text[eol] != '\n') eol++;
        std::string raw(text + pos, eol - pos);
        pos = eol + 1;
        lineno++;
        size_t cut = raw.find(";;");
        if (cut != std::string::npos) raw.resize(cut);
        std::string s = trim(raw);
        if (s.empty()) { if (eol >= len) break; continue; }
        std::vector<std::string> f = split(s);
        if (f.size() < 3) bad(VDL_ERR_PARSE, lineno, "expected '<id>,<Op>,...'");
        const std::string &op = f[1];
        Node n;
        n.line = lineno;
        n.field = "val";
        const bool named_output = op == "Output" && (f[0].empty() || !(isdigit((unsigned char)f[0][0]) || f[0][0] == '-'));
        if (named_output) { n.id = last_id + 1; n.field = f[0]; }
        else {
            int64_t id64 = to_int(f[0], lineno);
            if (id64 <= 0 || id64 > (1 << 24)) bad(VDL_ERR_PARSE, lineno, "statement id out of range");
            n.id = (int)id64;
        }
        last_id = n.id;
        if ((size_t)n.id >= P.nodes.size()) P.nodes.resize((size_t)n.id + 64);
        if (P.nodes[(size_t)n.id].id != 0) bad(VDL_ERR_PARSE, lineno, "Id " + std::to_string(n.id) + " defined twice");
        auto arity = [&](size_t k) {
            if (f.size() != k) bad(VDL_ERR_PARSE, lineno, op + " expects " + std::to_string(k) + " fields, got " + std::to_string(f.size()));
        };
        auto ref = [&](size_t k) { int r = to_ref(f[k], lineno); use(r, lineno); return r; };
        if (op == "Load") {
            arity(3); n.op = Op::Load; n.column = f[2];
            if (n.column.empty()) bad(VDL_ERR_PARSE, lineno, "Load: empty column name");
        } else if (op == "Project") { arity(3); n.op = Op::Project; n.a = ref(2); }
        else if (op == "Shuffle") { arity(3); n.op = Op::Shuffle; n.a = ref(2); }
        else if (op == "Semisort") { arity(3); n.op = Op::Semisort; n.a = ref(2); }
        else if (op == "RangeV") { arity(5); n.op = Op::RangeV; n.imm0 = to_int(f[2], lineno); n.a = ref(3); n.imm1 = to_int(f[4], lineno); }
        else if (op == "RangeC") {
            arity(5); n.op = Op::RangeC; n.imm0 = to_int(f[2], lineno); n.imm1 = to_int(f[3], lineno); n.imm2 = to_int(f[4], lineno);
            if (n.imm1 < 0) bad(VDL_ERR_PARSE, lineno, "RangeC: negative count");
        } else if (op == "Scatter") { arity(5); n.op = Op::Scatter; n.a = ref(2); n.b = ref(3); n.c = ref(4); }
        else if (op == "Like") {
            if (f.size() < 5) bad(VDL_ERR_PARSE, lineno, "Like expects 5 fields, got " + std::to_string(f.size()));
            n.op = Op::Like; n.a = ref(2); n.b = ref(3);
            size_t at = 0;
            for (int k = 0; k < 4; k++) at = s.find(',', at) + 1;
            n.pattern = s.substr(at);
            if (n.pattern.size() > 255) bad(VDL_ERR_UNSUPPORTED, lineno, "Like: pattern longer than 255 bytes");
        } else if (op == "Output") {
            if (f.size() != 3 && f.size() != 4) bad(VDL_ERR_PARSE, lineno, "Output expects 3 or 4 fields, got " + std::to_string(f.size()));
            n.op = Op::Materialize; n.a = ref(f.size() - 1);
        } else {
            arity(4);
            n.a = ref(2); n.b = ref(3);
            n.op = Op::Binary; n.bin = -1;
            for (int k = 0; k < B_COUNT; k++) if (op == kBinNames[k]) n.bin = k;
            if (n.bin < 0) {
                if (op == "FoldSelect") n.op = Op::FoldSelect;
                else if (op == "FoldSum") n.op = Op::FoldSum;
                else if (op == "FoldMin") n.op = Op::FoldMin;
                else if (op == "FoldMax") n.op = Op::FoldMax;
                else if (op == "FoldChoose") n.op = Op::FoldChoose;
                else if (op == "FoldCount") n.op = Op::FoldCount;
                else if (op == "Partition") n.op = Op::Partition;
                else if (op == "Gather") n.op = Op::Gather;
                else if (op == "CrossProductOuter") { n.op = Op::Cross; n.bin = 0; }
                else if (op == "CrossProductInner") { n.op = Op::Cross; n.bin = 1; }
                else bad(VDL_ERR_PARSE, lineno, "unknown operator '" + op + "'");
            }
        }
        P.nodes[(size_t)n.id] = n;
        P.order.push_back(n.id);
        if (n.op == Op::Materialize) P.outputs.push_back(n.id);
        if (eol >= len) break;
    }
    if (P.order.empty()) throw Error(VDL_ERR_PARSE, "empty program");
    return P;
}

Program parse_program(const char *text, size_t len) {
    if (std::string(text, len).find(",Output,") != std::string::npos) return parse_vlite_program(text, len);   // VLite dialect
    Program P;
    P.nodes.resize(64);
    int lineno = 0;
    size_t pos = 0;
    auto use = [&](int id, int line) -> const Node & {
        if (id <= 0 || (size_t)id >= P.nodes.size() || P.nodes[(size_t)id].id == 0)
            bad(VDL_ERR_PARSE, line, "reference to undefined vector Id " + std::to_string(id));
        return P.nodes[(size_t)id];
    };
    auto need_field = [&](const Node &v, const std::string &f, int line, const char *opn) {
        if (v.field != f)
            bad(VDL_ERR_SHAPE, line, std::string(opn) + ": operand Id " + std::to_string(v.id) + " has field '" + v.field +
                                         "', expected '" + f + "'");
    };
    while (pos <= len) {
        size_t eol = pos;
        while (eol < len && text[eol] != '\n') eol++;
        std::string raw(text + pos, eol - pos);
        pos = eol + 1;
        lineno++;
        size_t cut = raw.find(";;");
        if (cut != std::string::npos) raw.resize(cut);
        std::string s = trim(raw);
        if (s.empty()) { if (eol >= len) break; continue; }
        std::vector<std::string> f = split(s);
        if (f.size() < 2) bad(VDL_ERR_PARSE, lineno, "expected '<id>,<Op>,...'");
        int64_t id64 = to_int(f[0], lineno);
        if (id64 <= 0 || id64 > (1 << 24)) bad(VDL_ERR_PARSE, lineno, "statement id out of range");
        Node n;
        n.id = (int)id64;
        n.line = lineno;
        if ((size_t)n.id >= P.nodes.size()) P.nodes.resize((size_t)n.id + 64);
        if (P.nodes[(size_t)n.id].id != 0) bad(VDL_ERR_PARSE, lineno, "Id " + std::to_string(n.id) + " defined twice");
        const std::string &op = f[1];
        auto arity = [&](size_t k) {
            if (f.size() != k)
                bad(VDL_ERR_PARSE, lineno, op + " expects " + std::to_string(k) + " fields, got " + std::to_string(f.size()));
        };
        if (op == "Load") {
            arity(3);
            n.op = Op::Load;
            n.column = f[2];
            if (n.column.empty()) bad(VDL_ERR_PARSE, lineno, "Load: empty column name");
            // the struct field is the key path minus its first component (Vdl.hs:161-168)
            size_t dot = n.column.find('.');
            n.field = dot == std::string::npos ? n.column : n.column.substr(dot + 1);
        } else if (op == "Project") {           // Project,<out>,Id v,<in>  (Vdl.hs:422-423)
            arity(5);
            n.op = Op::Project;
            n.a = to_ref(f[3], lineno);
            need_field(use(n.a, lineno), f[4], lineno, "Project");
            n.field = f[2];
        } else if (op == "RangeV") {            // RangeV,val,<from>,Id v,<step>  (Vdl.hs:428-431)
            arity(6);
            n.op = Op::RangeV;
            n.imm0 = to_int(f[3], lineno);
            n.a = to_ref(f[4], lineno);
            n.imm1 = to_int(f[5], lineno);
            use(n.a, lineno);
            n.field = f[2];
        } else if (op == "RangeC") {            // RangeC,val,<from>,<count>,<step>  (Vdl.hs:433-434)
            arity(6);
            n.op = Op::RangeC;
            n.imm0 = to_int(f[3], lineno);
            n.imm1 = to_int(f[4], lineno);
            n.imm2 = to_int(f[5], lineno);
            if (n.imm1 < 0) bad(VDL_ERR_PARSE, lineno, "RangeC: negative count");
            n.field = f[2];
        } else if (op == "Gather") {            // Gather,Id src,Id pos,val  (Vdl.hs:438)
            arity(5);
            n.op = Op::Gather;
            n.a = to_ref(f[2], lineno);
            n.b = to_ref(f[3], lineno);
            need_field(use(n.b, lineno), f[4], lineno, "Gather");
            n.field = use(n.a, lineno).field;
        } else if (op == "Scatter") {           // Scatter,Id src,Id fold,val,Id pos,val  (Vdl.hs:441-442)
            arity(7);
            n.op = Op::Scatter;
            n.a = to_ref(f[2], lineno);
            n.b = to_ref(f[3], lineno);
            n.c = to_ref(f[5], lineno);
            need_field(use(n.b, lineno), f[4], lineno, "Scatter");
            need_field(use(n.c, lineno), f[6], lineno, "Scatter");
            n.field = use(n.a, lineno).field;
        } else if (op == "Shuffle") {           // Shuffle,Id v  (Vdl.hs:449-450)
            arity(3);
            n.op = Op::Shuffle;
            n.a = to_ref(f[2], lineno);
            n.field = use(n.a, lineno).field;
        } else if (op == "MaterializeCompact") {  // MaterializeCompact,Id v  (Vdl.hs:452-453)
            arity(3);
            n.op = Op::Materialize;
            n.a = to_ref(f[2], lineno);
            n.field = use(n.a, lineno).field;
        } else if (op == "Like") {                // Like,val,Id data,val,Id dict,val,<pattern>  (Vdl.hs:444-447)
            if (f.size() < 8) bad(VDL_ERR_PARSE, lineno, "Like expects 8 fields, got " + std::to_string(f.size()));
            n.op = Op::Like;
            n.a = to_ref(f[3], lineno);
            n.b = to_ref(f[5], lineno);
            need_field(use(n.a, lineno), f[4], lineno, "Like");
            need_field(use(n.b, lineno), f[6], lineno, "Like");
            size_t at = 0;                        // the pattern is everything after the 7th comma, commas included
            for (int k = 0; k < 7; k++) at = s.find(',', at) + 1;
            n.pattern = s.substr(at);
            if (n.pattern.size() > 255) bad(VDL_ERR_UNSUPPORTED, lineno, "Like: pattern longer than 255 bytes");
            n.field = f[2];
        } else if (op == "CrossProductOuter" || op == "CrossProductInner") {   // <op>,Id left,Id right  (Vdl.hs:412-416)
            arity(4);
            n.op = Op::Cross;
            n.bin = op == "CrossProductInner" ? 1 : 0;
            n.a = to_ref(f[2], lineno);
            n.b = to_ref(f[3], lineno);
            use(n.a, lineno); use(n.b, lineno);
            n.field = "val";
        } else if (op == "Semisort") {          // Semisort,Id v  (Vdl.hs:425-426; emitted for the VLite format only)
            arity(3);
            n.op = Op::Semisort;
            n.a = to_ref(f[2], lineno);
            use(n.a, lineno);
            n.field = "val";
        } else {
            // <BinOp|Fold|Partition>,val,Id a,val,Id b,val  (Vdl.hs:436-439)
            n.op = Op::Binary;
            n.bin = -1;
            for (int k = 0; k < B_COUNT; k++) if (op == kBinNames[k]) n.bin = k;
            if (n.bin < 0) {
                if (op == "FoldSelect") n.op = Op::FoldSelect;
                else if (op == "FoldSum") n.op = Op::FoldSum;
                else if (op == "FoldMin") n.op = Op::FoldMin;
                else if (op == "FoldMax") n.op = Op::FoldMax;
                else if (op == "FoldChoose") n.op = Op::FoldChoose;
                else if (op == "FoldCount") n.op = Op::FoldCount;
                else if (op == "Partition") n.op = Op::Partition;
                else bad(VDL_ERR_PARSE, lineno, "unknown operator '" + op + "'");
            }
            arity(7);
            n.a = to_ref(f[3], lineno);
            n.b = to_ref(f[5], lineno);
            need_field(use(n.a, lineno), f[4], lineno, op.c_str());
            need_field(use(n.b, lineno), f[6], lineno, op.c_str());
            n.field = f[2];
        }
        P.nodes[(size_t)n.id] = n;
        P.order.push_back(n.id);
        if (n.op == Op::Materialize) P.outputs.push_back(n.id);
        if (eol >= len) break;
    }
    if (P.order.empty()) throw Error(VDL_ERR_PARSE, "empty program");
    return P;
}

}  // namespace vdl
