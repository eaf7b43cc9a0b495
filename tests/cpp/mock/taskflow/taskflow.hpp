#pragma once
// Minimal single-threaded stand-in for the parts of cpp-taskflow the reference's coarse-space builders use
// (taskflow.emplace(callable).name(..), Task::succeed / precede, Executor::run(taskflow).get()).
#include <functional>
#include <memory>
#include <string>
#include <vector>
namespace tf {
struct Node {
  std::function<void()> fn;
  std::vector<Node*> after;   // must run before this node
  bool done = false;
  std::string name;
  void run()
  {
    if (done) return;
    for (auto* a : after) a->run();
    done = true;
    if (fn) fn();
  }
};
class Task {
public:
  Task() = default;
  explicit Task(Node* n) : n(n) {}
  Task& name(const std::string& s) { if (n) n->name = s; return *this; }
  template <class... T> Task& succeed(T&&... t) { (n->after.push_back(t.n), ...); return *this; }
  template <class... T> Task& precede(T&&... t) { (t.n->after.push_back(n), ...); return *this; }
  bool empty() const { return n == nullptr; }
  Node* n = nullptr;
};
class Taskflow {
public:
  explicit Taskflow(const std::string& = "") {}
  template <class F> Task emplace(F&& f)
  {
    nodes.push_back(std::make_unique<Node>());
    nodes.back()->fn = std::forward<F>(f);
    return Task(nodes.back().get());
  }
  std::vector<std::unique_ptr<Node>> nodes;
};
struct Future { void get() {} void wait() {} };
class Executor {
public:
  explicit Executor(int = 1) {}
  Future run(Taskflow& t) { for (auto& n : t.nodes) n->run(); return {}; }
};
}  // namespace tf
