//! `GpuRouter`: wraps `DefaultRouter` exactly like `rmqtt-cluster-broadcast/src/router.rs:22-62`
//! wraps it — every `Router` method is delegated to `inner`, except that `add`/`remove` are also
//! mirrored into the device table and `matches` runs on the GPU(s).
//!
//! * one `rgr_group` = one handle per device; filters and publishes are routed by `rgr_shard_assign`
//!   inside the library, so one device and eight devices are the same code here (`devices`);
//! * `matches` never touches the device itself: it enqueues into the deadline micro-batcher
//!   (`crate::batcher`) and awaits its slice of the batched pass — no lock is held across the wait
//!   and the blocking FFI call runs under `spawn_blocking`;
//! * pending `add`/`remove`s are committed (`rgr_group_commit`) by the batcher right before the next
//!   pass, on its blocking thread, so SUBSCRIBE never waits for the device;
//! * the device does the trie walk (`TopicTree::matches`, trie.rs:301-409) and answers, per matched filter in
//!   iteration order, with the sub id of that filter's first subscriber (`rgr_group_match_filter_subs`); the sub id
//!   leads to the filter string, and the per-client loop of `_matches` (router.rs:194-231: No Local by whole-`Id`
//!   equality, `$share` members, the v3/v5 collector) runs in the caller's task over `inner.relations`, the source
//!   of truth — N tokio workers expand N publishes in parallel, as in the reference;
//! * a sub id freed by `remove` is QUARANTINED until the next commit has dropped it from the device table, and every
//!   pass carries the mutation epoch it ran at (bumped by `remove` / `resync` only — an `add` never recycles an id that a
//!   pass in flight may hold): a publish whose pass is older than the last removal is matched again on its own, so a
//!   recycled id can never resolve to a relation the device did not match (round-2 / round-3 advisor findings);
//! * `$share` members (router.rs:202-213) are collected per (filter, group) while one filter's relations go by
//!   and `SharedSubscription::choice` picks one (router.rs:236-255) — same place, same arguments.
//!
//! Source only (no rustc in the build image).  The C++ twin that IS compiled and tested against the
//! oracle's `DefaultRouter` — same structure, same order of decisions, incl. the `$share` path, the
//! sharded table and the batcher — is rmqtt_amd/host/gpu_router.{hpp,cpp} (tests/test_host_router.py).
use std::ffi::CStr;
use std::sync::atomic::{AtomicBool, Ordering};
use std::sync::{Arc, RwLock};
use std::time::Duration;

use ahash::AHashMap as HashMap;
use async_trait::async_trait;
use rmqtt::context::ServerContext;
use rmqtt::router::{DefaultRouter, Router};
use rmqtt::types::*;
use rmqtt::utils::Counter;
use rmqtt::Result;

use crate::batcher::{match_many, Batcher, GroupPtr, MutationEpoch};
use crate::ffi::*;

struct Group(*mut rgr_group);
unsafe impl Send for Group {}
unsafe impl Sync for Group {} // the C ABI is thread-safe (include/rmqtt_gpu_router.h, "threading")
impl Drop for Group {
    fn drop(&mut self) {
        unsafe { rgr_group_destroy(self.0) }
    }
}

/// Dense u32 ids with reference counts (owner id per `Id`, client index per (node, ClientId)).
#[derive(Default)]
struct Dense<K: std::hash::Hash + Eq + Clone> {
    ids: HashMap<K, (u32, u32)>, // key -> (id, refs)
    free: Vec<u32>,
    next: u32,
}
impl<K: std::hash::Hash + Eq + Clone> Dense<K> {
    fn acquire(&mut self, k: &K) -> u32 {
        if let Some(e) = self.ids.get_mut(k) {
            e.1 += 1;
            return e.0;
        }
        let id = self.free.pop().unwrap_or_else(|| { self.next += 1; self.next - 1 });
        self.ids.insert(k.clone(), (id, 1));
        id
    }
    fn release(&mut self, k: &K) {
        if let Some(e) = self.ids.get_mut(k) {
            e.1 -= 1;
            if e.1 == 0 {
                let id = e.0;
                self.ids.remove(k);
                self.free.push(id);
            }
        }
    }
    fn find(&self, k: &K) -> u32 {
        self.ids.get(k).map(|e| e.0).unwrap_or(RGR_ID_NONE)
    }
}

/// sub_id slab: dense relation id -> (filter, client).  `inner.relations` stays the source of truth
/// for `Id` and options; the slab only says which relation a hit is.
#[derive(Default)]
struct Slab {
    slots: Vec<Option<(TopicFilter, ClientId, Id)>>,
    free: Vec<u32>,
    /// sub ids freed since the last commit: the device table may still hold them, so they are not handed out again
    /// before a commit has gone by (then they move to `free`)
    quarantine: Vec<u32>,
    ids: HashMap<(TopicFilter, ClientId), u32>,
    per_filter: HashMap<TopicFilter, usize>,
    owners: Dense<Id>,
    clients: Dense<(NodeId, ClientId)>,
    nodes: Vec<NodeId>,
    node_idx: HashMap<NodeId, u16>,
}

/// Read guard over the sub id slab for `GpuShared`: sub id -> (filter, client, Id) of the relation.
pub(crate) struct SlabRead<'a>(std::sync::RwLockReadGuard<'a, Slab>);
impl SlabRead<'_> {
    pub(crate) fn relation(&self, sub_id: u32) -> Option<&(TopicFilter, ClientId, Id)> {
        self.0.slots.get(sub_id as usize).and_then(|x| x.as_ref())
    }
}

#[derive(Clone)]
pub struct GpuRouter {
    scx: ServerContext,
    inner: DefaultRouter,
    g: Arc<Group>,
    slab: Arc<RwLock<Slab>>,
    dirty: Arc<AtomicBool>,
    epoch: Arc<MutationEpoch>,
    /// serialises [epoch read + commit] of every pass and of `match_one_exclusive`; lock order: commit_lock, then `slab`
    commit_lock: Arc<std::sync::Mutex<()>>,
    batcher: Arc<Batcher>,
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(rgr_last_error()).to_string_lossy().into_owned() }
}

fn flags(opts: &SubscriptionOptions) -> u8 {
    let mut f = 0;
    if !opts.is_v3() { f |= RGR_SUB_V5; }
    if opts.no_local() == Some(true) { f |= RGR_SUB_NO_LOCAL; }
    if opts.retain_as_published() == Some(true) { f |= RGR_SUB_RAP; }
    if opts.has_shared_group() { f |= RGR_SUB_SHARED; }
    f
}

impl GpuRouter {
    /// `devices`: one shard per HIP device ordinal (a single entry = one GPU).  Must be called inside the
    /// tokio runtime (the batcher's driver task is spawned here).
    pub fn new(scx: ServerContext, devices: &[i32], max_batch: usize, max_delay: Duration) -> Result<Self> {
        let cfg = rgr_config::default();
        let mut g = std::ptr::null_mut();
        if unsafe { rgr_group_create(&cfg, devices.as_ptr(), devices.len() as u32, &mut g) } != RGR_OK {
            return Err(anyhow::anyhow!("rgr_group_create: {}", last_error()));
        }
        let g = Arc::new(Group(g));
        let dirty = Arc::new(AtomicBool::new(false));
        let epoch = Arc::new(MutationEpoch::default());
        let slab = Arc::new(RwLock::new(Slab::default()));
        let commit_lock = Arc::new(std::sync::Mutex::new(()));
        let commit = Self::committer(GroupPtr(g.0), dirty.clone(), epoch.clone(), slab.clone(), commit_lock.clone());
        // pending subscription changes become visible right before the next pass, on the batcher's blocking thread
        let batcher = Batcher::spawn(GroupPtr(g.0), max_batch, max_delay, commit);
        Ok(Self { inner: DefaultRouter::new(Some(scx.clone())), scx, g, slab, dirty, epoch, commit_lock, batcher: Arc::new(batcher) })
    }

    /// What runs right before a device pass: the mutation epoch the pass may claim (read BEFORE the commit, so a
    /// mutation racing with the commit makes the pass look older than it is, never newer), then `rgr_group_commit`;
    /// the sub ids freed before this commit leave quarantine once it succeeded.
    fn committer(gp: GroupPtr, dirty: Arc<AtomicBool>, epoch: Arc<MutationEpoch>, slab: Arc<RwLock<Slab>>, commit_lock: Arc<std::sync::Mutex<()>>)
                 -> impl Fn() -> std::result::Result<u64, String> + Send + Sync + Clone + 'static {
        // Several passes may be in flight (batcher.rs MAX_IN_FLIGHT): the epoch read and the commit it may trigger are ONE critical
        // section, so a pass can only start on the table of a finished commit.  Without it pass B could skip the commit that pass A
        // is still running (dirty already swapped), walk the previous device epoch, and come back with a removed sub id after A's
        // commit has released that id to a new subscription — under an unchanged mutation epoch.
        move || {
            let _one_at_a_time = commit_lock.lock().unwrap();
            let e = epoch.get();
            if dirty.swap(false, Ordering::AcqRel) {
                let released = std::mem::take(&mut slab.write().unwrap().quarantine);
                if unsafe { rgr_group_commit(gp.0) } != RGR_OK {
                    dirty.store(true, Ordering::Release);
                    slab.write().unwrap().quarantine.extend(released);
                    return Err(format!("rgr_group_commit: {}", last_error()));
                }
                slab.write().unwrap().free.extend(released);
            }
            Ok(e)
        }
    }

    /// sub id of each matched filter's first subscriber -> the filter strings, in `TopicTree::matches` order.
    /// `None`: an id of the pass does not resolve.  Under an equal epoch that cannot happen; if it ever does the publish is
    /// matched again (callers) instead of silently losing that filter's subscribers (round-3 advisor).
    fn filters_of(s: &Slab, first_subs: &[u32]) -> Option<Vec<TopicFilter>> {
        first_subs.iter().map(|sid| s.slots.get(*sid as usize).and_then(|x| x.as_ref()).map(|(f, _, _)| f.clone())).collect()
    }

    /// Commit, match ONE topic and resolve its filters while holding the slab exclusively (blocking: call under `spawn_blocking`).
    fn match_one_exclusive(&self, topic: &str) -> std::result::Result<Vec<TopicFilter>, String> {
        let _one_at_a_time = self.commit_lock.lock().unwrap();
        let mut s = self.slab.write().unwrap();
        if self.dirty.swap(false, Ordering::AcqRel) {
            if unsafe { rgr_group_commit(self.g.0) } != RGR_OK {
                self.dirty.store(true, Ordering::Release);
                return Err(format!("rgr_group_commit: {}", last_error()));
            }
            let released = std::mem::take(&mut s.quarantine);
            s.free.extend(released);
        }
        let hits = unsafe { match_many(GroupPtr(self.g.0), &[topic.to_string()], self.epoch.get()) }?.remove(0)?;
        // committed and matched with the slab held: every id resolves, or the table and the slab disagree — an error, not a drop
        Self::filters_of(&s, &hits.first_subs).ok_or_else(|| format!("device table out of step with the sub id slab for topic `{topic}`"))
    }

    pub fn _inner(&self) -> &DefaultRouter {
        &self.inner
    }

    // ---- what `crate::shared::GpuShared` (the consumer of the delivery stage) needs from the router ---------------------------------
    pub(crate) fn group_ptr(&self) -> GroupPtr { GroupPtr(self.g.0) }
    /// The same "epoch, then commit what is pending" step the match batcher runs before a pass (see `committer`).
    pub(crate) fn pass_committer(&self) -> impl Fn() -> std::result::Result<u64, String> + Send + Sync + Clone + 'static {
        Self::committer(GroupPtr(self.g.0), self.dirty.clone(), self.epoch.clone(), self.slab.clone(), self.commit_lock.clone())
    }
    pub(crate) fn mutation_epoch(&self) -> u64 { self.epoch.get() }
    pub(crate) fn slab_read(&self) -> SlabRead<'_> { SlabRead(self.slab.read().unwrap()) }
    /// Dense id of the publisher's `Id` (No Local compares whole Ids, router.rs:198); RGR_ID_NONE when it holds no subscription.
    pub(crate) fn owner_id_of(&self, id: &Id) -> u32 { self.slab.read().unwrap().owners.find(id) }
    /// The relation's v5 subscription identifier, from the source of truth (`inner.relations`).
    pub(crate) fn subscription_identifier_of(&self, filter: &TopicFilter, client_id: &ClientId) -> Option<SubscriptionIdentifier> {
        self.inner.relations.get(filter).and_then(|rels| rels.get(client_id).and_then(|(_, opts)| opts.subscription_identifier()))
    }

    /// Mirror one relation into the device table (after `inner.add` accepted it).
    fn mirror_add(&self, topic_filter: &str, id: &Id, opts: &SubscriptionOptions) -> Result<()> {
        let mut s = self.slab.write().unwrap();
        let key = (TopicFilter::from(topic_filter), id.client_id.clone());
        let sub_id = match s.ids.get(&key) {
            Some(sid) => {
                // re-subscribe (HashMap::insert replaces, router.rs:447): same relation id, new Id / options
                let sid = *sid;
                if let Some((_, _, old_id)) = s.slots[sid as usize].take() {
                    s.owners.release(&old_id);
                    s.clients.release(&(old_id.node_id, old_id.client_id.clone()));
                }
                sid
            }
            None => {
                let sid = s.free.pop().unwrap_or_else(|| { s.slots.push(None); (s.slots.len() - 1) as u32 });
                s.ids.insert(key.clone(), sid);
                *s.per_filter.entry(key.0.clone()).or_default() += 1;
                sid
            }
        };
        let owner_id = s.owners.acquire(id);
        let client_idx = s.clients.acquire(&(id.node_id, id.client_id.clone()));
        let node_idx = match s.node_idx.get(&id.node_id) {
            Some(i) => *i,
            None => {
                if s.nodes.len() >= 0xFFFF { return Err(anyhow::anyhow!("more than 65535 distinct node ids")); }
                let i = s.nodes.len() as u16;
                s.nodes.push(id.node_id);
                s.node_idx.insert(id.node_id, i);
                i
            }
        };
        s.slots[sub_id as usize] = Some((key.0, key.1, id.clone()));
        let rc = unsafe {
            rgr_group_subscribe_ex(self.g.0, topic_filter.as_ptr() as _, topic_filter.len() as u32, sub_id, opts.qos_value(), flags(opts),
                                   node_idx, owner_id, client_idx)
        };
        if rc != RGR_OK { return Err(anyhow::anyhow!("rgr_group_subscribe_ex: {}", last_error())); }
        // NO epoch bump: an add cannot make a sub id of a pass in flight resolve to another relation — ids are only recycled
        // out of the quarantine that `mirror_remove` fills, and that bumps.  (Bumping here sent every batched publish through
        // the exclusive re-match path under plain subscribe churn: round-3 advisor.)
        self.dirty.store(true, Ordering::Release);
        Ok(())
    }

    fn mirror_remove(&self, topic_filter: &str, id: &Id) -> Result<()> {
        let mut s = self.slab.write().unwrap();
        let key = (TopicFilter::from(topic_filter), id.client_id.clone());
        let Some(sid) = s.ids.remove(&key) else { return Ok(()) };
        if let Some((_, _, old_id)) = s.slots[sid as usize].take() {
            s.owners.release(&old_id);
            s.clients.release(&(old_id.node_id, old_id.client_id.clone()));
        }
        s.quarantine.push(sid); // not `free`: the device keeps the id until the next commit
        let last = {
            let n = s.per_filter.get_mut(&key.0).map(|n| { *n -= 1; *n }).unwrap_or(0);
            if n == 0 { s.per_filter.remove(&key.0); }
            n == 0 // the filter leaves the trie with its last relation (router.rs:484-490)
        };
        let rc = unsafe { rgr_group_unsubscribe(self.g.0, topic_filter.as_ptr() as _, topic_filter.len() as u32, sid, last as i32) };
        if rc != RGR_OK { return Err(anyhow::anyhow!("rgr_group_unsubscribe: {}", last_error())); }
        // dirty BEFORE the bump (round-3 advisor): the committer reads the epoch first and `dirty` second, so a pass that
        // claims the bumped epoch is guaranteed to have seen dirty == true and committed the removal.  The other order let a
        // pass claim epoch E+1 while the device still answered with the removed id.
        self.dirty.store(true, Ordering::Release);
        self.epoch.bump();
        Ok(())
    }

    /// Rebuild the device table from `inner.relations` with ONE bulk call — the restore path of the cluster routers
    /// (rmqtt-cluster-raft/src/router.rs:557-566 re-inserts every relation after a snapshot; the snapshot itself —
    /// postcard + zstd/lz4 of `relations`, :387-463 — is decoded by the broker as today, `rgr_group_subscribe_bulk`
    /// then builds the trie in sorted order instead of 10 M single inserts).  `restore` writes `inner.relations`
    /// directly, without `add` calls, so this is the one call a maintainer adds at its end; whatever was mirrored
    /// before leaves the device table first (the C++ mirror's `GpuRouter::restore` swaps in a fresh group instead).
    pub fn resync(&self) -> Result<()> {
        let mut s = self.slab.write().unwrap();
        let old: Vec<(u32, TopicFilter)> =
            s.slots.iter().enumerate().filter_map(|(i, x)| x.as_ref().map(|(f, _, _)| (i as u32, f.clone()))).collect();
        for (sid, f) in old {
            let left = s.per_filter.get_mut(&f).map(|n| { *n -= 1; *n }).unwrap_or(0);
            let rc = unsafe { rgr_group_unsubscribe(self.g.0, f.as_ptr() as _, f.len() as u32, sid, (left == 0) as i32) };
            if rc != RGR_OK { return Err(anyhow::anyhow!("rgr_group_unsubscribe: {}", last_error())); }
        }
        *s = Slab::default();
        let (mut blob, mut offs) = (Vec::<u8>::new(), vec![0u64]);
        let (mut sub_ids, mut qos, mut fl) = (Vec::<u32>::new(), Vec::<u8>::new(), Vec::<u8>::new());
        let (mut owners, mut clients) = (Vec::<u32>::new(), Vec::<u32>::new());
        for e in self.inner.relations.iter() {
            let filter = e.key().clone();
            for (client_id, (id, opts)) in e.value().iter() {
                let sid = s.slots.len() as u32;
                s.slots.push(Some((filter.clone(), client_id.clone(), id.clone())));
                s.ids.insert((filter.clone(), client_id.clone()), sid);
                *s.per_filter.entry(filter.clone()).or_default() += 1;
                owners.push(s.owners.acquire(id));
                clients.push(s.clients.acquire(&(id.node_id, client_id.clone())));
                blob.extend_from_slice(filter.as_bytes());
                offs.push(blob.len() as u64);
                sub_ids.push(sid);
                qos.push(opts.qos_value());
                fl.push(flags(opts));
            }
        }
        let mut rejected = 0u64;
        let n = sub_ids.len() as u64;
        let rc = unsafe { rgr_group_subscribe_bulk(self.g.0, blob.as_ptr(), offs.as_ptr(), n, sub_ids.as_ptr(), qos.as_ptr(), fl.as_ptr(), &mut rejected) };
        if rc != RGR_OK || rejected != 0 { return Err(anyhow::anyhow!("rgr_group_subscribe_bulk: rc {rc}, {rejected} filters rejected: {}", last_error())); }
        if unsafe { rgr_group_sub_attrs_bulk(self.g.0, sub_ids.as_ptr(), owners.as_ptr(), clients.as_ptr(), n) } != RGR_OK {
            return Err(anyhow::anyhow!("rgr_group_sub_attrs_bulk: {}", last_error()));
        }
        self.dirty.store(true, Ordering::Release); // (before the bump: see mirror_remove)
        self.epoch.bump();
        Ok(())
    }
}

#[async_trait]
impl Router for GpuRouter {
    async fn add(&self, topic_filter: &str, id: Id, opts: SubscriptionOptions) -> Result<()> {
        self.inner.add(topic_filter, id.clone(), opts.clone()).await?; // rmqtt/src/router.rs:434-453 (rejects invalid filters)
        self.mirror_add(topic_filter, &id, &opts)
    }

    async fn remove(&self, topic_filter: &str, id: Id) -> Result<bool> {
        let removed = self.inner.remove(topic_filter, id.clone()).await?; // rmqtt/src/router.rs:456-496
        if removed {
            self.mirror_remove(topic_filter, &id)?;
        }
        Ok(removed)
    }

    /// rmqtt/src/router.rs:499-501 / 174-265.
    async fn matches(&self, this_id: Id, topic: &TopicName) -> Result<SubRelationsMap> {
        // the device pass (trie walk), shared with every concurrent publish: one sub id per matched filter
        let hits = self.batcher.matches(topic).await.map_err(|e| anyhow::anyhow!(e))?;
        // sub id -> filter string, under the slab's read lock (never held across an await).  The answer is only used when no
        // add / remove has happened since the pass: otherwise this publish is matched again on its own (rare).
        let resolved: Option<Vec<TopicFilter>> = {
            let s = self.slab.read().unwrap();
            if self.epoch.get() == hits.epoch { Self::filters_of(&s, &hits.first_subs) } else { None }
        };
        let filters = match resolved {
            Some(f) => f,
            None => {
                // the table changed since the batched pass: this publish is matched again on its own with the slab held exclusively
                // from the commit to the resolution (mirror_add / mirror_remove wait for it) — always current, no retry loop
                let (this, t) = (self.clone(), topic.to_string());
                tokio::task::spawn_blocking(move || this.match_one_exclusive(&t))
                    .await
                    .map_err(|e| anyhow::anyhow!(e.to_string()))?
                    .map_err(|e| anyhow::anyhow!(e))?
            }
        };

        let mut collector_map: SubscriptioRelationsCollectorMap = Default::default();
        type Member = (NodeId, ClientId, SubscriptionOptions, Option<Vec<SubscriptionIdentifier>>, Option<IsOnline>);
        for filter in filters.iter() {
            // Id and options come from the source of truth; a filter whose relations are gone since the pass is skipped,
            // as the reference would no longer see them either (router.rs:184)
            let members: Vec<(ClientId, Id, SubscriptionOptions)> = match self.inner.relations.get(filter) {
                Some(rels) => rels.iter().map(|(c, (id, o))| (c.clone(), id.clone(), o.clone())).collect(),
                None => continue,
            };
            let mut groups: HashMap<SharedGroup, Vec<Member>> = Default::default(); // router.rs:183-192, one filter at a time
            for (client_id, id, opts) in members {
                if opts.no_local() == Some(true) && id == this_id { continue; } // router.rs:196-201: whole-`Id` equality
                if let Some(group) = opts.shared_group() {
                    // router.rs:204-213
                    let online = self.is_online(id.node_id, &client_id).await;
                    groups.entry(group.clone()).or_default().push((id.node_id, client_id, opts.clone(), None, Some(online)));
                    continue;
                }
                collector_map.entry(id.node_id).or_default().add(filter, client_id, opts, None); // router.rs:214-229
            }
            // select a subscriber from every shared group of this filter (router.rs:236-255)
            for (group, mut s_subs) in groups.drain() {
                let group_cids = s_subs.iter().map(|(_, cid, _, _, _)| cid.clone()).collect();
                if let Some((idx, is_online)) = self.scx.extends.shared_subscription().await.choice(&self.scx, &group, &this_id, topic, &s_subs).await {
                    let (node_id, client_id, opts, _, _) = s_subs.remove(idx);
                    collector_map.entry(node_id).or_default().add(filter, client_id, opts, Some((group, is_online, group_cids)));
                }
            }
        }
        Ok(collector_map.into_iter().map(|(n, c)| (n, c.into())).collect()) // router.rs:258-261
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
