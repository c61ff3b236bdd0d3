// csv_simd.hpp — C++ host-side mirror of the reference crate's public surface for the stage-1
// path: StructureError, StructureIndex, Header, TapeCore/Tape, Boundary/boundaries, Chunk and
// the RecordSource accessors.  Same names, argument meaning and error behaviour as the Rust
// (reference src/error.rs, src/stage1.rs:60-76, src/tape.rs, src/record_source.rs); the stage-1
// scan itself (reader::read, src/reader.rs:150) runs on the GPU behind include/csvsimd.h.
//
// Header-only, scalar host code — exactly as in the reference, where everything after stage 1
// is O(1) / first-line work on the CPU.
#pragma once
#include <cstdint>
#include <optional>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace csv_simd {

// reference src/error.rs:7-21
enum class StructureError : int {
    Ok = 0,
    Io = -1,
    MissingValue = -2,
    InvalidState = -3,
    InvalidCsvFormat = -4,
};

// reference src/stage1.rs:470-480 (only the two variants the crate constructs)
enum class NewLine { LF = 0, CRLF = 1 };

// reference src/stage1.rs:60-76: StructureIndex(Vec<CodeUnitPos>), CodeUnitPos(usize),
// KeyToPos(usize) — transparent usize newtypes, kept as plain integers here.
using CodeUnitPos = std::uint64_t;
using KeyToPos = std::uint64_t;

// non-owning view of the stage-1 result (tape[0] == 0 sentinel, then ascending offsets)
struct StructureIndex {
    const CodeUnitPos* data = nullptr;
    std::uint64_t len_ = 0;
    std::uint64_t len() const { return len_; }
    CodeUnitPos operator[](std::uint64_t i) const { return data[i]; }
};

// reference src/tape.rs:217-277
struct Header {
    std::vector<std::string> header;
    NewLine new_line = NewLine::LF;
    std::uint32_t field_cnt = 0;
    std::uint8_t delimiter = 0x2C;
    std::uint32_t record_offset = 0;

    // Header::new (src/tape.rs:226-273).  The Rust indexes memmap[header_end_idx + 1] unchecked
    // by length and panics when the file has no byte after the first line end; that panic is
    // reported here as InvalidState instead of aborting the process.
    static StructureError create(const std::uint8_t* bytes, std::uint64_t len, Header& out) {
        std::uint64_t end = 0;
        while (end < len && bytes[end] != 0x0d && bytes[end] != 0x0a) ++end;
        if (end + 1 >= len) return StructureError::InvalidState;
        out.new_line = bytes[end + 1] == 0x0a ? NewLine::CRLF : NewLine::LF;
        std::uint64_t start = 0;  // skip the byte-order mark: any run of ef/bb/bf
        while (start < len && (bytes[start] == 0xef || bytes[start] == 0xbb || bytes[start] == 0xbf)) ++start;
        if (start > end) return StructureError::InvalidState;  // Rust: slice start > end panics
        out.header.clear();
        std::string_view line(reinterpret_cast<const char*>(bytes) + start, end - start);
        std::size_t pos = 0;
        for (;;) {  // str::split(",") yields one item more than there are commas
            const std::size_t comma = line.find(',', pos);
            std::string_view name = line.substr(pos, comma == std::string_view::npos ? line.size() - pos : comma - pos);
            out.header.emplace_back(trim(name));
            if (comma == std::string_view::npos) break;
            pos = comma + 1;
        }
        out.field_cnt = static_cast<std::uint32_t>(out.header.size());
        out.delimiter = 0x2C;
        out.record_offset = static_cast<std::uint32_t>(end);
        return StructureError::Ok;
    }

  private:
    static std::string_view trim(std::string_view s) {  // str::trim on ASCII white space
        auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
        while (!s.empty() && ws(s.front())) s.remove_prefix(1);
        while (!s.empty() && ws(s.back())) s.remove_suffix(1);
        return s;
    }
};

// reference src/tape.rs:281-284, 385-428
struct Boundary {
    std::uint64_t start, len;
    bool operator==(const Boundary& o) const { return start == o.start && len == o.len; }
};

inline std::optional<std::vector<Boundary>> boundaries(std::uint32_t task_size, std::uint8_t job_count) {
    if (task_size == 0 || job_count == 0) return std::nullopt;
    if (task_size < job_count) return std::vector<Boundary>{Boundary{0, task_size}};
    const std::uint32_t job_size = task_size / job_count;
    const std::uint32_t remainder = task_size % job_count;
    std::vector<Boundary> out;
    out.reserve(job_count);
    std::uint32_t acc_end = 0, share_remainder = 1;
    for (std::uint32_t i = 0; i < job_count; ++i) {
        if (share_remainder == 1 && i >= remainder) share_remainder = 0;
        out.push_back(Boundary{acc_end, job_size + share_remainder});
        acc_end += job_size + share_remainder;
    }
    return out;
}

// reference src/tape.rs:13-19
struct Chunk {
    std::uint8_t id;
    KeyToPos start, end;
    std::uint32_t record_cnt;
};

// reference src/record_source.rs:68-147: the default methods of trait RecordSource, written
// against the five accessors a source provides.
template <class Source>
struct RecordSource {
    using Span = std::pair<std::uint64_t, std::uint64_t>;  // bytes[first, second)

    // Ok(None) -> returns Ok with an empty optional
    static StructureError seek_record(const Source& s, std::uint32_t record_idx, std::optional<Span>& out) {
        out.reset();
        const auto rc = s.record_cnt();
        if (!rc) return StructureError::InvalidState;
        if (record_idx + 1 >= *rc) return StructureError::Ok;
        const auto jump = s.record_jump_size();
        if (!jump) return StructureError::InvalidState;
        const std::uint32_t field_cnt = s.field_cnt();
        const std::uint32_t idx_start = (record_idx + 1) * static_cast<std::uint32_t>(*jump);
        const CodeUnitPos mem_start = s.index()[idx_start];
        const CodeUnitPos mem_end = s.index()[static_cast<std::uint64_t>(idx_start) + field_cnt];
        out = Span{mem_start + 1, mem_end};
        return StructureError::Ok;
    }
    static StructureError seek_field(const Source& s, std::uint32_t record_idx, std::uint32_t field_idx,
                                     std::optional<Span>& out) {
        out.reset();
        const auto rc = s.record_cnt();
        if (!rc) return StructureError::InvalidState;
        if (record_idx + 1 >= *rc) return StructureError::Ok;
        if (field_idx >= s.field_cnt()) return StructureError::Ok;
        const std::uint32_t field_cnt = s.field_cnt();
        const std::uint32_t row_size = s.new_line_tag() == NewLine::CRLF ? field_cnt + 1 : field_cnt;
        const std::uint32_t idx_start = (record_idx + 1) * row_size + field_idx;
        const CodeUnitPos mem_start = s.index()[idx_start];
        const CodeUnitPos mem_end = s.index()[static_cast<std::uint64_t>(idx_start) + 1];
        out = Span{mem_start + 1, mem_end};
        return StructureError::Ok;
    }
};

// reference src/tape.rs:185-212, 301-347 (TapeCore) and :74-153 (Tape).  One class: `init`
// is what turns a core into a usable tape, as Tape::from_core does.
class Tape {
  public:
    Header header;
    std::uint32_t record_cnt_ = 0;
    KeyToPos record_jump_size_ = 0;

    // TapeCore::create + Tape::from_core (src/lib.rs:68-69)
    static StructureError from_core(const std::uint8_t* bytes, std::uint64_t len, StructureIndex index,
                                    Header header, Tape& out) {
        out.bytes_ = bytes;
        out.len_ = len;
        out.index_ = index;
        out.header = std::move(header);
        return out.init();
    }

    // Tape::chunks (src/tape.rs:95-140)
    StructureError chunks(std::uint8_t num, std::vector<Chunk>& out) const {
        const auto b = boundaries(record_cnt_, num);
        if (!b) return StructureError::InvalidState;
        out.clear();
        std::uint8_t id = 0;
        for (const Boundary& x : *b)
            out.push_back(Chunk{id++, x.start * record_jump_size_, (x.start + x.len) * record_jump_size_,
                                static_cast<std::uint32_t>(x.len)});
        out[0].start = record_jump_size_;  // skip the header row
        out[0].record_cnt -= 1;
        return StructureError::Ok;
    }

    // accessors of impl RecordSource for &Tape (src/tape.rs:155-174)
    std::optional<std::uint32_t> record_cnt() const { return record_cnt_; }
    const StructureIndex& index() const { return index_; }
    std::optional<KeyToPos> record_jump_size() const { return record_jump_size_; }
    std::uint32_t field_cnt() const { return header.field_cnt; }
    NewLine new_line_tag() const { return header.new_line; }
    const std::uint8_t* data_bytes() const { return bytes_; }
    std::uint64_t data_len() const { return len_; }

  private:
    // TapeCore::init (src/tape.rs:315-347)
    StructureError init() {
        if (index_.len() == 0) return StructureError::InvalidState;  // Rust: len()-1 underflow panic
        record_jump_size_ = header.new_line == NewLine::CRLF ? header.field_cnt + 1ull : header.field_cnt;
        if (record_jump_size_ == 0) return StructureError::InvalidState;  // Rust: division by zero panic
        record_cnt_ = static_cast<std::uint32_t>((index_.len() - 1) / record_jump_size_);
        const std::uint64_t problem = (index_.len() - 1) % record_jump_size_;
        if (problem != 0) return StructureError::InvalidCsvFormat;
        return StructureError::Ok;
    }
    const std::uint8_t* bytes_ = nullptr;
    std::uint64_t len_ = 0;
    StructureIndex index_;
};

}  // namespace csv_simd
