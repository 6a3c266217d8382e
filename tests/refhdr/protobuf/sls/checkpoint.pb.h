// tests/refhdr -- TEST INFRASTRUCTURE: stand-in for the protoc-generated checkpoint.pb.h (core/protobuf/sls/checkpoint.proto is in the
// reference tree, its generated header is not): the accessors core/file_server/checkpoint/RangeCheckpoint.h uses, declarations only.
#pragma once
#include <cstdint>
#include <string>

namespace logtail {
class RangeCheckpointPB {
public:
    void set_committed(bool);
    bool committed() const;
    uint64_t sequence_id() const;
    void set_sequence_id(uint64_t);
    bool has_hash_key() const;
    const std::string& hash_key() const;
    void set_hash_key(const std::string&);
    uint64_t read_offset() const;
    void set_read_offset(uint64_t);
    uint64_t read_length() const;
    void set_read_length(uint64_t);
};
}  // namespace logtail
