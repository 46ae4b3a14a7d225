// stand-in for ros `filters` <filters/filter_base.h>: FilterBase<T> with configure()/update()/getParam()
#pragma once
#include <map>
#include <string>

namespace filters {

struct ParamValue {
  enum Kind { kDouble, kInt, kString } kind = kDouble;
  double d = 0.0;
  int i = 0;
  std::string s;
  ParamValue() {}
  ParamValue(double v) : kind(kDouble), d(v) {}
  ParamValue(int v) : kind(kInt), d(v), i(v) {}
  ParamValue(const char* v) : kind(kString), s(v) {}
  ParamValue(const std::string& v) : kind(kString), s(v) {}
};
typedef std::map<std::string, ParamValue> ParamMap;

template <typename T>
class FilterBase {
 public:
  FilterBase() : configured_(false) {}
  virtual ~FilterBase() {}
  // the real class is configured from the parameter server; the stand-in takes the parameter map directly
  bool configure(const std::string& name, const ParamMap& params) {
    filter_name_ = name;
    params_ = params;
    configured_ = configure();
    return configured_;
  }
  virtual bool update(const T& data_in, T& data_out) = 0;
  const std::string& getName() const { return filter_name_; }

 protected:
  virtual bool configure() = 0;
  bool getParam(const std::string& name, double& value) const {
    ParamMap::const_iterator it = params_.find(name);
    if (it == params_.end() || it->second.kind == ParamValue::kString) return false;
    value = it->second.d;
    return true;
  }
  bool getParam(const std::string& name, int& value) const {
    ParamMap::const_iterator it = params_.find(name);
    if (it == params_.end() || it->second.kind != ParamValue::kInt) return false;
    value = it->second.i;
    return true;
  }
  bool getParam(const std::string& name, std::string& value) const {
    ParamMap::const_iterator it = params_.find(name);
    if (it == params_.end() || it->second.kind != ParamValue::kString) return false;
    value = it->second.s;
    return true;
  }
  bool configured_;
  std::string filter_name_;
  ParamMap params_;
};

}  // namespace filters
