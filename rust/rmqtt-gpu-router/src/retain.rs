//! `GpuRetainStorage`: the `RetainStorage` trait (rmqtt/src/retain.rs:100-186) with the RetainTree's
//! wildcard query (`RetainTree::matches`, retain.rs:450-526) on the GPU.
//!
//! The messages stay on the host in a slab keyed by a dense `topic_id`; the device holds only the trie of
//! retained topic NAMES (`rgr_retain_topic_add` / `_remove`) and answers a SUBSCRIBE filter with the ids of
//! the matching topics (`rgr_retain_match_ranges`: ranges of the host-mirrored preorder value array).  This is the in-memory storage (`DefaultRetainStorage`,
//! retain.rs:200-330); the retainer plugin's KV variants do the same around their store — match first, then one
//! KV get per matched topic (rmqtt-plugins/rmqtt-retainer/src/storage.rs:576-641) — and can call
//! `GpuRetainIndex::query` in place of their `RetainTree` lookup (storage.rs:604-611).
//! The retained set runs in two tiers (`retain_delta_max`): a new retained topic recompiles a small delta
//! table instead of the whole set.  Installed at the slot the retainer uses (rmqtt-retainer/src/lib.rs:191).
//!
//! Source only (no rustc in the build image); C++ twin, compiled and tested against the oracle's RetainTree:
//! rmqtt_amd/host/gpu_retain.{hpp,cpp} (tests/test_host_router.py::test_retain_storage_mirror).
use std::ffi::CStr;
use std::str::FromStr;
use std::sync::atomic::{AtomicBool, Ordering};
use std::sync::Mutex;
use std::time::Duration;

use ahash::AHashMap as HashMap;
use async_trait::async_trait;
use rmqtt::retain::RetainStorage;
use rmqtt::topic::Topic;
use rmqtt::types::*;
use rmqtt::utils::Counter;
use rmqtt::Result;

use crate::ffi::*;

struct Handle(*mut rgr_handle);
unsafe impl Send for Handle {}
unsafe impl Sync for Handle {}
impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { rgr_destroy(self.0) }
    }
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(rgr_last_error()).to_string_lossy().into_owned() }
}

/// The topic-name index alone: what `RetainTree<V>` is to its users, with `V` = a dense id.
pub struct GpuRetainIndex {
    h: Handle,
    dirty: AtomicBool,
}

impl GpuRetainIndex {
    pub fn new(device: i32, retain_delta_max: u32) -> Result<Self> {
        let cfg = rgr_config { device, retain_delta_max, ..Default::default() };
        let mut h = std::ptr::null_mut();
        if unsafe { rgr_create(&cfg, &mut h) } != RGR_OK {
            return Err(anyhow::anyhow!("rgr_create: {}", last_error()));
        }
        Ok(Self { h: Handle(h), dirty: AtomicBool::new(false) })
    }

    /// RetainTree::insert (retain.rs:373-386): Err = `Topic::from_str` failed.
    pub fn insert(&self, topic: &str, id: u32) -> Result<()> {
        if unsafe { rgr_retain_topic_add(self.h.0, topic.as_ptr() as _, topic.len() as u32, id) } != RGR_OK {
            return Err(anyhow::anyhow!("invalid topic `{topic}`: {}", last_error()));
        }
        self.dirty.store(true, Ordering::Release);
        Ok(())
    }

    /// RetainTree::remove (retain.rs:393-413)
    pub fn remove(&self, topic: &str) -> bool {
        let ok = unsafe { rgr_retain_topic_remove(self.h.0, topic.as_ptr() as _, topic.len() as u32) } == RGR_OK;
        if ok { self.dirty.store(true, Ordering::Release); }
        ok
    }

    /// RetainTree::matches for a batch of filters: per filter `Err` (invalid filter) or the ids of the matching
    /// topics.  Blocking (device pass): call it under `spawn_blocking`.
    pub fn query(&self, filters: &[&str]) -> Result<Vec<std::result::Result<Vec<u32>, String>>> {
        if self.dirty.swap(false, Ordering::AcqRel) && unsafe { rgr_retain_commit(self.h.0) } != RGR_OK {
            self.dirty.store(true, Ordering::Release);
            return Err(anyhow::anyhow!("rgr_retain_commit: {}", last_error()));
        }
        let mut blob = Vec::new();
        let mut offs = vec![0u64];
        for f in filters {
            blob.extend_from_slice(f.as_bytes());
            offs.push(blob.len() as u64);
        }
        // the dense answer: per filter a short list of RANGES of the preorder value array, which the library mirrors on the host —
        // `a/#` is one range.  16 bytes per range cross PCIe instead of 12 per hit (r4: rgr_retain_match_batch shipped tuples).
        let mut res: rgr_retain_ranges = unsafe { std::mem::zeroed() };
        if unsafe { rgr_retain_match_ranges(self.h.0, blob.as_ptr(), offs.as_ptr(), filters.len() as u32, &mut res) } != RGR_OK {
            return Err(anyhow::anyhow!("rgr_retain_match_ranges: {}", last_error()));
        }
        let out = unsafe {
            let status = std::slice::from_raw_parts(res.status, filters.len());
            let ro = std::slice::from_raw_parts(res.range_offsets, filters.len() + 1);
            let ranges = if res.n_ranges == 0 { &[][..] } else { std::slice::from_raw_parts(res.ranges, res.n_ranges as usize) };
            let vals: [&[rgr_retain_val]; 2] = [0, 1].map(|t| {
                if res.vals[t].is_null() { &[][..] } else { std::slice::from_raw_parts(res.vals[t], res.n_vals[t] as usize) }
            });
            (0..filters.len())
                .map(|i| {
                    if status[i] != RGR_TOPIC_OK {
                        return Err(format!("invalid topic filter `{}`", filters[i]));
                    }
                    let mut ids = Vec::new();
                    for r in &ranges[ro[i] as usize..ro[i + 1] as usize] {
                        let (tier, len) = ((r.len >> 31) as usize, (r.len & 0x7fff_ffff) as usize);
                        // entries removed / replaced since the base tier was compiled carry RGR_RETAIN_HIT_DEAD: skipped
                        ids.extend(vals[tier][r.begin as usize..r.begin as usize + len].iter().filter(|v| v.flags & RGR_RETAIN_HIT_DEAD == 0).map(|v| v.topic_id));
                    }
                    Ok(ids)
                })
                .collect()
        };
        unsafe { rgr_retain_ranges_free(&mut res) };
        Ok(out)
    }
}

#[derive(Default)]
struct Messages {
    slab: Vec<Option<(TopicName, TimedValue<Retain>)>>,
    /// `clock` value at which the slot was last given to a topic (parallel to `slab`); `clock` counts assignments
    stamp: Vec<u64>,
    clock: u64,
    free: Vec<u32>,
    ids: HashMap<TopicName, u32>,
}

pub struct GpuRetainStorage {
    index: std::sync::Arc<GpuRetainIndex>,
    messages: std::sync::Arc<Mutex<Messages>>,
    retaineds: Counter,
}

impl GpuRetainStorage {
    pub fn new(device: i32) -> Result<Self> {
        Ok(Self { index: std::sync::Arc::new(GpuRetainIndex::new(device, 65536)?), messages: Default::default(), retaineds: Counter::new() })
    }

    /// retain.rs:216-226 — `RetainTree::retain(usize::MAX, |tv| !tv.is_expired())`
    pub async fn remove_expired_messages(&self) -> usize {
        let mut m = self.messages.lock().unwrap();
        let mut removed = 0;
        for id in 0..m.slab.len() {
            let expired = matches!(&m.slab[id], Some((_, tv)) if tv.is_expired());
            if !expired { continue; }
            let (topic, _) = m.slab[id].take().unwrap();
            self.index.remove(&topic);
            m.ids.remove(&topic);
            m.free.push(id as u32);
            self.retaineds.dec();
            removed += 1;
        }
        removed
    }
}

#[async_trait]
impl RetainStorage for GpuRetainStorage {
    fn enable(&self) -> bool { true }

    /// retain.rs:229-247 (`set_with_timeout`): remove the old value; a non-empty payload stores the new one.
    async fn set(&self, topic: &TopicName, retain: Retain, expiry_interval: Option<Duration>) -> Result<()> {
        let mut m = self.messages.lock().unwrap();
        let had = m.ids.get(topic).copied();
        if retain.publish.payload.is_empty() {
            match had {
                Some(id) => {
                    self.index.remove(topic);
                    m.slab[id as usize] = None;
                    m.free.push(id);
                    m.ids.remove(topic);
                    self.retaineds.dec();
                }
                // still has to be a valid topic name (`Topic::from_str(topic)?`, retain.rs:235)
                None => { Topic::from_str(topic)?; }
            }
            return Ok(());
        }
        let id = had.unwrap_or_else(|| m.free.pop().unwrap_or_else(|| { m.slab.push(None); m.stamp.push(0); (m.slab.len() - 1) as u32 }));
        if let Err(e) = self.index.insert(topic, id) {
            if had.is_none() { m.free.push(id); }
            return Err(e);
        }
        m.slab[id as usize] = Some((topic.clone(), TimedValue::new(retain, expiry_interval)));
        if had.is_none() {
            m.clock += 1;
            let now = m.clock;
            m.stamp[id as usize] = now;
            m.ids.insert(topic.clone(), id);
            self.retaineds.inc();
        }
        Ok(())
    }

    /// retain.rs:250-267 (`get_message`): match on the device, drop expired entries, clone the messages.
    async fn get(&self, topic_filter: &TopicFilter) -> Result<Vec<(TopicName, Retain)>> {
        let (index, filter) = (self.index.clone(), topic_filter.to_string());
        let asked_at = self.messages.lock().unwrap().clock;
        let ids = tokio::task::spawn_blocking(move || index.query(&[filter.as_str()])).await??;
        let ids = ids.into_iter().next().unwrap().map_err(|e| anyhow::anyhow!(e))?;
        let m = self.messages.lock().unwrap();
        // a topic id freed and re-issued between the device query and this lookup would name ANOTHER topic: every slot
        // carries the stamp of its last assignment, and a slot assigned after the query started is not part of its answer
        Ok(ids
            .into_iter()
            .filter(|id| m.stamp.get(*id as usize).map(|s| *s <= asked_at).unwrap_or(false))
            .filter_map(|id| m.slab.get(id as usize).and_then(|e| e.as_ref()))
            .filter(|(_, tv)| !tv.is_expired())
            .map(|(t, tv)| (t.clone(), tv.value().clone()))
            .collect())
    }

    async fn count(&self) -> isize { self.retaineds.count() }
    async fn max(&self) -> isize { self.retaineds.max() }
}
