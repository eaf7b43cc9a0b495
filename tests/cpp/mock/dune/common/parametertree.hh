#pragma once
#include <map>
#include <sstream>
#include <string>
#include "exceptions.hh"
namespace Dune {
class ParameterTree {
public:
  bool hasKey(const std::string& k) const { return values.count(k) > 0; }
  bool hasSub(const std::string& k) const { return subs.count(k) > 0; }
  std::string& operator[](const std::string& k) { return values[k]; }
  ParameterTree& sub(const std::string& k) { return subs[k]; }
  const ParameterTree& sub(const std::string& k) const
  {
    static const ParameterTree empty;
    auto it = subs.find(k);
    return it == subs.end() ? empty : it->second;
  }
  template <class T>
  T get(const std::string& k, const T& def) const
  {
    auto it = values.find(k);
    if (it == values.end()) return def;
    std::istringstream s(it->second);
    T v;
    s >> v;
    return v;
  }
  std::string get(const std::string& k, const char* def) const { return get<std::string>(k, std::string(def)); }
private:
  std::map<std::string, std::string> values;
  std::map<std::string, ParameterTree> subs;
};
}  // namespace Dune
