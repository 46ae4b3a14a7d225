// stand-in for pluginlib's class loader registry (in-container test build only)
#include <pluginlib/class_list_macros.h>

namespace pluginlib_stub {
std::map<std::string, Factory>& registry() {
  static std::map<std::string, Factory> r;
  return r;
}
}  // namespace pluginlib_stub
