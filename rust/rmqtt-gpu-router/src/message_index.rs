//! `GpuMessageIndex`: the topic index of rmqtt-message-storage on the device.
//!
//! `RamMessageManagerInner::topic_tree: RwLock<RetainTree<MsgID>>` (rmqtt-plugins/rmqtt-message-storage/src/ram.rs:158) is the same
//! structure and query shape as the retained-message tree, so it binds the same `rgr_retain_*` calls through `GpuRetainIndex`:
//!
//! * `_set` (ram.rs:333-334, :351 / :361): `Topic::from_str(&publish.topic)` + `push(Level::Normal(msg_id.to_string()))`, then
//!   `topic_tree.insert(&topic, msg_id)`  ->  `set(topic, msg_id)`;
//! * expiry sweep (ram.rs:226-233): the same topic with the same id level, `topic_tree.remove(&topic)`  ->  `remove(topic, msg_id)`;
//! * `_get` (ram.rs:381-394): `Topic::from_str(topic_filter)`, `push(Level::SingleWildcard)` unless the last level is `#`,
//!   `topic_tree.matches(&topic)` -> msg ids  ->  `get(topic_filter)`;
//! * `values_size` / `nodes_size` (ram.rs:460-461) for the metrics.
//!
//! What a maintainer changes in ram.rs: the field's type (`topic_tree: GpuMessageIndex`) and the four call sites above — the stored
//! messages (`messages` / `messages_encode`), the expiries heap and the `forwardeds` map stay where they are.  The index gains one
//! leaf per stored message, so it runs in the library's two-tier mode (additions recompile a small delta table, removals set dead
//! bits, the tiers merge every `retain_delta_max` additions).  The device pass is blocking: `get` is called under `spawn_blocking`
//! (the reference already treats this lookup as potentially slow, ram.rs:396-405).
//!
//! Source only (no rustc in the build image).  The C++ twin that is compiled and tested against the oracle's `RetainTree` with the
//! same set / get / remove sequences is rmqtt_amd/host/gpu_retain.{hpp,cpp} (`GpuMessageIndex`; tests/test_host_router.py).
use std::collections::HashMap;
use std::sync::Mutex;

use rmqtt::types::MsgID;
use rmqtt::Result;

use crate::retain::GpuRetainIndex;

/// one new leaf per stored message: merge the tiers every this many additions (C++ twin: `kDefaultDeltaMax`)
pub const DEFAULT_DELTA_MAX: u32 = 65536;

#[derive(Default)]
struct Ids {
    slab: Vec<MsgID>,             // dense topic id -> MsgID
    free: Vec<u32>,
    of: HashMap<MsgID, u32>,
}

pub struct GpuMessageIndex {
    index: GpuRetainIndex,
    ids: Mutex<Ids>,
}

/// `Topic::push(Level::Normal(msg_id.to_string()))` in string form: one more '/'-separated level.
fn with_msg_level(topic: &str, msg_id: MsgID) -> String {
    format!("{topic}/{msg_id}")
}

impl GpuMessageIndex {
    pub fn new(device: i32) -> Result<Self> {
        Ok(Self { index: GpuRetainIndex::new(device, DEFAULT_DELTA_MAX)?, ids: Mutex::new(Ids::default()) })
    }

    /// ram.rs:333-334 + :351 / :361.  Err: `Topic::from_str` failed.
    pub fn set(&self, topic: &str, msg_id: MsgID) -> Result<()> {
        let mut ids = self.ids.lock().unwrap();
        let (id, had) = match ids.of.get(&msg_id) {
            Some(id) => (*id, true),
            None => match ids.free.pop() {
                Some(id) => (id, false),
                None => { ids.slab.push(0); ((ids.slab.len() - 1) as u32, false) }
            },
        };
        if let Err(e) = self.index.insert(&with_msg_level(topic, msg_id), id) {
            if !had { ids.free.push(id); }
            return Err(e);
        }
        ids.slab[id as usize] = msg_id;
        if !had { ids.of.insert(msg_id, id); }
        Ok(())
    }

    /// ram.rs:226-233.  false: nothing was stored there.
    pub fn remove(&self, topic: &str, msg_id: MsgID) -> bool {
        let mut ids = self.ids.lock().unwrap();
        if !self.index.remove(&with_msg_level(topic, msg_id)) {
            return false;
        }
        if let Some(id) = ids.of.remove(&msg_id) {
            ids.free.push(id);
        }
        true
    }

    /// ram.rs:381-394: the ids of the messages whose topic matches the filter (ascending).  Blocking (device pass).
    pub fn get(&self, topic_filter: &str) -> Result<Vec<MsgID>> {
        // last level `#`  <=>  the string is "#" or ends in "/#" (levels are the '/'-separated pieces)
        let multi = topic_filter == "#" || topic_filter.ends_with("/#");
        let q = if multi { topic_filter.to_string() } else { format!("{topic_filter}/+") };
        let ids = self.ids.lock().unwrap();          // (held across the query like the reference's read guard: a removal cannot recycle an id under it)
        let hits = self.index.query(&[q.as_str()])?.remove(0).map_err(|e| anyhow::anyhow!(e))?;
        let mut out: Vec<MsgID> = hits.into_iter().map(|id| ids.slab[id as usize]).collect();
        out.sort_unstable();
        Ok(out)
    }

    /// ram.rs:460: `topic_tree.values_size()`
    pub fn values_size(&self) -> usize {
        self.ids.lock().unwrap().of.len()
    }
}
