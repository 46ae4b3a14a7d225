// stand-in for <pluginlib/class_list_macros.h>: registers a factory the test driver can look up by type name
#pragma once
#include <map>
#include <string>

namespace pluginlib_stub {
typedef void* (*Factory)();
std::map<std::string, Factory>& registry();  // defined once, in stubs/pluginlib/registry.cpp (part of the test build)
struct Registrar {
  Registrar(const char* name, Factory f) { registry()[name] = f; }
};
}  // namespace pluginlib_stub

#define PLUGINLIB_STUB_CAT2(a, b) a##b
#define PLUGINLIB_STUB_CAT(a, b) PLUGINLIB_STUB_CAT2(a, b)
#define PLUGINLIB_EXPORT_CLASS(class_type, base_class_type)                                           \
  namespace {                                                                                         \
  void* PLUGINLIB_STUB_CAT(pluginlib_stub_make_, __LINE__)() { return static_cast<base_class_type*>(new class_type()); } \
  pluginlib_stub::Registrar PLUGINLIB_STUB_CAT(pluginlib_stub_reg_, __LINE__)(#class_type,            \
                                                                            &PLUGINLIB_STUB_CAT(pluginlib_stub_make_, __LINE__)); \
  }
