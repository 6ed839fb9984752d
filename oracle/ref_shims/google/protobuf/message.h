// DEPENDENCY SHIM (oracle/_ref build only): enough of protobuf for voxblox's Layer/Block
// headers to parse.  Nothing on the integration path serialises.
#pragma once
#include <cstdint>
#include <string>
namespace google { namespace protobuf {
class MessageLite { public: virtual ~MessageLite() {} };
class Message : public MessageLite {};
namespace io {
class ZeroCopyInputStream {}; class ZeroCopyOutputStream {};
class IstreamInputStream : public ZeroCopyInputStream { public: template <typename T> explicit IstreamInputStream(T*) {} };
class OstreamOutputStream : public ZeroCopyOutputStream { public: template <typename T> explicit OstreamOutputStream(T*) {} };
class CodedInputStream { public: template <typename T> explicit CodedInputStream(T*) {} };
class CodedOutputStream { public: template <typename T> explicit CodedOutputStream(T*) {} };
}}}  // namespace google::protobuf::io
