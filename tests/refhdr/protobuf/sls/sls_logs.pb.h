// tests/refhdr -- TEST INFRASTRUCTURE: stand-in for the protoc-generated sls_logs.pb.h (core/protobuf/sls/sls_logs.proto is in the
// reference tree, its generated header is not): the types core/app_config/AppConfig.h and core/common/TimeUtil.h name, declarations only.
#pragma once
#include <string>
namespace sls_logs {
class Log;
class Log_Content;
class LogGroup;
class LogTag {
public:
    const std::string& key() const;
    const std::string& value() const;
    void set_key(const std::string&);
    void set_value(const std::string&);
};
}  // namespace sls_logs
