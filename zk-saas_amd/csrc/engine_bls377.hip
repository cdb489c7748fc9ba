// Engine instantiation for one curve (separate translation unit so the curves compile in parallel).
#include "curves.hpp"
#include "engine_impl.hpp"
namespace zk {
IEngine* make_engine_bls377(int l, int device) {
  auto* e = new Engine<CfgBls377>(l, device);
  if (e->init() != ZK_OK) {
    // keep the object: the caller reads last error and destroys it
  }
  return e;
}
}  // namespace zk
