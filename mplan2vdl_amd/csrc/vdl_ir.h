// vdl_ir.h -- in-memory form of a VDL program (one Node per text line).
// Grammar and operand order: /root/reference/src/Vdl.hs:410-477 (toVoodooList, printLine).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "vdl_scan_desc.h"

namespace vdl {

enum class Op : int {
    Load, Project, RangeV, RangeC, Binary, FoldSelect, FoldSum, FoldMin, FoldMax, FoldChoose,
    FoldCount, Gather, Scatter, Partition, Shuffle, Materialize, Like, Cross, Semisort
};

extern const char *const kBinNames[B_COUNT];
const char *op_name(Op op, int bin);

struct Node {
    int id = 0;
    Op op = Op::Load;
    int bin = -1;                // BinOp when op == Binary
    int a = -1, b = -1, c = -1;  // operand ids in print order
    int64_t imm0 = 0, imm1 = 0, imm2 = 0;  // RangeV: from, step; RangeC: from, count, step
    std::string column;          // Load: key path
    std::string pattern;         // Like: SQL LIKE pattern ('%', '_'), no escape character
    std::string field;           // struct field this vector's data lives in after the op
    int line = 0;
};

struct Program {
    std::vector<Node> nodes;     // indexed by id (slot 0 unused; undefined ids have id == 0)
    std::vector<int> order;      // ids in text order (def-before-use is enforced)
    std::vector<int> outputs;    // MaterializeCompact ids in text order
    const Node &at(int id) const { return nodes[(size_t)id]; }
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

Program parse_program(const char *text, size_t len);

// int64 semantics shared by constant folding on the host and the device kernels
// (wrap-around; x/0 = x%0 = 0; BitShift sign encodes direction, Vlite.hs:205-208).
#if defined(__HIPCC__)
#define VDL_HD __host__ __device__
#else
#define VDL_HD
#endif
VDL_HD inline int64_t apply_bin(int op, int64_t a, int64_t b) {
    switch (op) {
    case B_LAND: return (a != 0) && (b != 0);
    case B_LOR:  return (a != 0) || (b != 0);
    case B_BAND: return a & b;
    case B_BOR:  return a | b;
    case B_SHIFT:
        if (b >= 0) return a >> (b > 63 ? 63 : b);
        if (b <= -64) return 0;
        return (int64_t)((uint64_t)a << (unsigned)(-b));
    case B_EQ:  return a == b;
    case B_ADD: return (int64_t)((uint64_t)a + (uint64_t)b);
    case B_SUB: return (int64_t)((uint64_t)a - (uint64_t)b);
    case B_GT:  return a > b;
    case B_MUL: return (int64_t)((uint64_t)a * (uint64_t)b);
    case B_DIV:
        if (b == 0) return 0;
        if (b == -1) return (int64_t)(0 - (uint64_t)a);
        return a / b;
    default:
        if (b == 0 || b == -1) return 0;
        return a % b;
    }
}

}  // namespace vdl
