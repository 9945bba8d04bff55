// RetainTree twin (rmqtt/src/retain.rs) — host table + device snapshot state.
#pragma once
namespace rgr {
struct RetainState {};
}  // namespace rgr
