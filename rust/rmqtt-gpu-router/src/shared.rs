//! `GpuShared`: the consumer of the device's delivery stage — wraps `DefaultShared` the way
//! `rmqtt-cluster-broadcast/src/shared.rs` (`ClusterShared { inner: DefaultShared, .. }`) wraps it and is installed through
//! `*scx.extends.shared_mut().await = Box::new(..)` (rmqtt/src/extend.rs:123; rmqtt-cluster-broadcast/src/lib.rs:141).  Every
//! `Shared` method is delegated to `inner` except `forwards`:
//!
//! * reference (`rmqtt/src/shared.rs:735-820`): `router.matches(from.id, &publish.topic)` builds a `SubRelationsMap` (per hit: clones
//!   of the filter, the client id and the options into per-node collectors, router.rs:214-229), `forwards_to` (:876-963) walks this
//!   node's relations — per relation `publish.clone()`, Retain-As-Published, `qos.less_value`, subscription ids, `self.tx(&client_id)`,
//!   `tx.unbounded_send(Message::Forward(from, p))`;
//! * here: the publish joins a deadline micro-batch (`DeliverBatcher`, the delivery-kind twin of `crate::batcher`), ONE device pass per
//!   batch (`rgr_group_match_batch_deliver` with the publishes' real qos / retain and the publishers' owner ids) takes every per-hit
//!   decision — No Local by whole-`Id` equality (router.rs:196-201), the v5 collector's first hit per client (types.rs:524-539),
//!   qos' = min, retain' — and every publish's hits go from `(sub_id, delivery word)` straight into the sessions' channels.  No
//!   `SubRelationsMap`, no per-hit clones of filter strings and options; what is left per recipient is the session lookup and the send.
//! * a publish the device cannot finish takes `inner.forwards` unchanged: `target_clientid` publishes (shared.rs:744: no matching),
//!   publishes whose hits include `$share` members (`SharedSubscription::choice` is the broker's, router.rs:236-255), and publishes
//!   whose pass is older than the last removal (a recycled sub id; same epoch rule as `GpuRouter::matches`).
//!
//! Source only (no rustc in the build image).  The C++ twin that IS compiled, tested against the oracle's `forwards` dump and timed
//! (`bench.py --router-e2e`: 13.0 k publishes/s = 193 M recipients/s at BASELINE configs[2] against 4.8 k/s for the reference-shaped CPU
//! pass on 256 threads) is rmqtt_amd/host/gpu_shared.{hpp,cpp} (tests/test_host_router.py).
use std::sync::Arc;
use std::time::Duration;

use async_trait::async_trait;
use rmqtt::context::ServerContext;
use rmqtt::shared::{DefaultShared, Entry, Shared};
use rmqtt::types::*;
use rmqtt::Result;
use tokio::sync::{mpsc, oneshot, Semaphore};

use crate::batcher::{GroupPtr, MAX_IN_FLIGHT};
use crate::ffi::*;
use crate::router::GpuRouter;

/// One publish's share of a delivery pass: `(sub_id, delivery word)` per hit in `TopicTree::matches` order, and the mutation
/// epoch the pass ran at.  `Err`: `Topic::from_str` failed (router.rs:177) or the pass failed.
pub struct DeliverHits {
    pub hits: Vec<(u32, u32)>,
    pub epoch: u64,
}
struct DeliverRequest {
    topic: TopicName,
    from_owner: u32, // owner id of the publisher's `Id` (RGR_ID_NONE: holds no subscription)
    qos_retain: u32,
    reply: oneshot::Sender<std::result::Result<DeliverHits, String>>,
}

/// The delivery-kind twin of `crate::batcher::Batcher`: same deadline batching, the pass is `rgr_group_match_batch_deliver`.
pub struct DeliverBatcher {
    tx: mpsc::UnboundedSender<DeliverRequest>,
}

impl DeliverBatcher {
    pub fn spawn<F>(g: GroupPtr, max_batch: usize, max_delay: Duration, before_pass: F) -> Self
    where
        F: Fn() -> std::result::Result<u64, String> + Send + Sync + 'static,
    {
        let (tx, mut rx) = mpsc::unbounded_channel::<DeliverRequest>();
        let before_pass = Arc::new(before_pass);
        let in_flight = Arc::new(Semaphore::new(MAX_IN_FLIGHT));
        tokio::spawn(async move {
            while let Some(first) = rx.recv().await {
                let mut reqs = vec![first];
                let deadline = tokio::time::sleep(max_delay);
                tokio::pin!(deadline);
                while reqs.len() < max_batch {
                    tokio::select! {
                        _ = &mut deadline => break,
                        r = rx.recv() => match r { Some(r) => reqs.push(r), None => break },
                    }
                }
                let bp = before_pass.clone();
                let Ok(permit) = in_flight.clone().acquire_owned().await else { break };
                tokio::spawn(async move {
                    let work: Vec<(TopicName, u32, u32)> = reqs.iter().map(|r| (r.topic.clone(), r.from_owner, r.qos_retain)).collect();
                    let res = tokio::task::spawn_blocking(move || {
                        let epoch = bp()?;
                        unsafe { deliver_many(g, &work, epoch) }
                    })
                    .await;
                    drop(permit);
                    match res {
                        Ok(Ok(per_publish)) => reqs.into_iter().zip(per_publish).for_each(|(r, h)| { let _ = r.reply.send(h); }),
                        Ok(Err(e)) => reqs.into_iter().for_each(|r| { let _ = r.reply.send(Err(e.clone())); }),
                        Err(e) => reqs.into_iter().for_each(|r| { let _ = r.reply.send(Err(e.to_string())); }),
                    }
                });
            }
        });
        Self { tx }
    }

    pub async fn deliver(&self, topic: &TopicName, from_owner: u32, qos_retain: u32) -> std::result::Result<DeliverHits, String> {
        let (reply, rx) = oneshot::channel();
        self.tx.send(DeliverRequest { topic: topic.clone(), from_owner, qos_retain, reply }).map_err(|e| e.to_string())?;
        rx.await.map_err(|e| e.to_string())?
    }
}

/// One delivery pass (`rgr_group_match_batch_deliver`): 12-byte tuples whose third word is the delivery word.
unsafe fn deliver_many(g: GroupPtr, work: &[(TopicName, u32, u32)], epoch: u64) -> std::result::Result<Vec<std::result::Result<DeliverHits, String>>, String> {
    let mut blob = Vec::new();
    let mut offs = vec![0u64];
    let mut attrs = Vec::with_capacity(work.len());
    for (t, from, qr) in work {
        blob.extend_from_slice(t.as_bytes());
        offs.push(blob.len() as u64);
        attrs.push(rgr_publish_attr { from_id: *from, qos_retain: *qr });
    }
    let mut res: rgr_result = std::mem::zeroed();
    if rgr_group_match_batch_deliver(g.0, blob.as_ptr(), offs.as_ptr(), work.len() as u32, attrs.as_ptr(), &mut res) != RGR_OK {
        return Err(std::ffi::CStr::from_ptr(rgr_last_error()).to_string_lossy().into_owned());
    }
    let status = std::slice::from_raw_parts(res.status, work.len());
    let ho = std::slice::from_raw_parts(res.hit_offsets, work.len() + 1);
    let tuples = if res.n_hits == 0 { &[][..] } else { std::slice::from_raw_parts(res.tuples, res.n_hits as usize) };
    let out = (0..work.len())
        .map(|i| {
            if status[i] != RGR_TOPIC_OK {
                Err(format!("invalid topic `{}`", work[i].0))
            } else {
                Ok(DeliverHits { hits: tuples[ho[i] as usize..ho[i + 1] as usize].iter().map(|t| (t.sub_id, t.qos_flags)).collect(), epoch })
            }
        })
        .collect();
    rgr_result_free(&mut res);
    Ok(out)
}

#[derive(Clone)]
pub struct GpuShared {
    inner: DefaultShared,
    scx: ServerContext,
    router: GpuRouter,
    batcher: Arc<DeliverBatcher>,
}

impl GpuShared {
    /// Must be called inside the tokio runtime, after `GpuRouter::new` (the batcher commits the router's pending changes before a pass).
    pub fn new(scx: ServerContext, router: GpuRouter, max_batch: usize, max_delay: Duration) -> Self {
        let batcher = DeliverBatcher::spawn(router.group_ptr(), max_batch, max_delay, router.pass_committer());
        Self { inner: DefaultShared::new(Some(scx.clone())), scx, router, batcher: Arc::new(batcher) }
    }

    /// `forwards_to` (shared.rs:876-963) from delivery words.  `Err(stale)`: the publish cannot be finished from these hits — `stale` = a removal overtook
    /// the pass (it joins another batch), otherwise `$share` members are among the hits (the reference's own path).
    fn deliver(&self, from: &From, publish: &Publish, hits: &DeliverHits) -> std::result::Result<(ForwardedRecipients, Vec<(To, From, Publish, Reason)>), bool> {
        let this_node = self.scx.node.id();
        let mut ok: ForwardedRecipients = Vec::new();
        let mut errs = Vec::new();
        // v5 rows wait until the publish's hits have all gone by: a client's FIRST hit also carries the subscription identifiers of
        // its later (duplicate) hits (types.rs:526-534); v3 rows, then the v5 collector's, is the order `From<Collector>` yields too
        let mut rows5: Vec<(ClientId, QoS, bool, Vec<SubscriptionIdentifier>)> = Vec::new();
        let mut index5: Option<ahash::AHashMap<ClientId, usize>> = None;
        let mut send = |client_id: ClientId, qos: QoS, retain: bool, sub_ids: Option<Vec<SubscriptionIdentifier>>| {
            let mut p = publish.clone(); // shared.rs:899-908 with the device's qos' / retain'
            p.dup = false;
            p.retain = retain;
            p.qos = qos;
            p.packet_id = None;
            if let (Some(ids), Some(props)) = (sub_ids, p.properties.as_mut()) {
                props.subscription_ids = ids;
            }
            let Some((tx, to)) = self.inner.tx(&client_id) else {
                errs.push((To::from(0, client_id), from.clone(), p, Reason::from_static("the client has disconnected")));
                return;
            };
            match tx.unbounded_send(Message::Forward(from.clone(), p)) {
                Ok(()) => ok.push((client_id, None)),
                Err(e) => {
                    if let Message::Forward(from, p) = e.into_inner() {
                        errs.push((to, from, p, Reason::from_static("Connection Tx is closed")));
                    }
                }
            }
        };
        {
            let slab = self.router.slab_read();
            if self.router.mutation_epoch() != hits.epoch {
                return Err(true); // a removal since the pass: a sub id may have been recycled
            }
            for (sub_id, w) in hits.hits.iter().copied() {
                if (w >> 8) & (RGR_SUB_SHARED as u32) != 0 {
                    return Err(false); // $share members: SharedSubscription::choice is the broker's
                }
                if w & RGR_HIT_NO_LOCAL != 0 {
                    continue; // router.rs:196-201, decided on the device
                }
                let Some((filter, client_id, id)) = slab.relation(sub_id) else { return Err(true) };
                if id.node_id != this_node {
                    continue; // shared.rs:809-815: a single-node Shared only warns about other nodes' relations
                }
                let qos = QoS::try_from((w & RGR_HIT_QOS_MASK) as u8).unwrap_or(QoS::AtMostOnce);
                let retain = w & RGR_HIT_RETAIN != 0;
                let v5 = (w >> 8) & (RGR_SUB_V5 as u32) != 0;
                let ident = if v5 { self.router.subscription_identifier_of(filter, client_id) } else { None };
                if w & RGR_HIT_V5_DUP != 0 {
                    if let Some(ident) = ident {
                        let idx = index5.get_or_insert_with(|| rows5.iter().enumerate().map(|(i, r)| (r.0.clone(), i)).collect());
                        if let Some(i) = idx.get(client_id) {
                            rows5[*i].3.push(ident);
                        }
                    }
                    continue;
                }
                if v5 {
                    if let Some(idx) = index5.as_mut() {
                        idx.insert(client_id.clone(), rows5.len());
                    }
                    rows5.push((client_id.clone(), qos, retain, ident.into_iter().collect()));
                } else {
                    send(client_id.clone(), qos, retain, None);
                }
            }
        }
        for (client_id, qos, retain, ids) in rows5 {
            send(client_id, qos, retain, if ids.is_empty() { None } else { Some(ids) });
        }
        Ok((ok, errs))
    }
}

#[async_trait]
impl Shared for GpuShared {
    fn entry(&self, id: Id) -> Box<dyn Entry> { self.inner.entry(id) }
    fn exist(&self, client_id: &str) -> bool { self.inner.exist(client_id) }

    /// rmqtt/src/shared.rs:735-820
    async fn forwards(
        &self,
        msg_id: Option<MsgID>,
        from: From,
        publish: Publish,
    ) -> std::result::Result<ForwardedCount, (ForwardedCount, Vec<(To, From, Publish, Reason)>)> {
        if publish.target_clientid.is_some() {
            return self.inner.forwards(msg_id, from, publish).await; // shared.rs:744-770: no matching involved
        }
        let qos_retain = (publish.qos.value() as u32 & 3) | if publish.retain { 4 } else { 0 };
        // A publish whose pass a removal overtook joins another batch (a few times) instead of taking the reference's path, which would run a device pass
        // of ONE publish through `Router::matches`: measured on the C++ twin, two removals a second took Shared::forwards from 4.1 M to 1.5 M publishes/s
        // that way (profiles/r07x_*).  (The C++ twin goes further — rmqtt_amd/host/gpu_router.hpp `limbo_`: its delivery passes survive removals.)
        let mut tries = 0;
        let (recipients, errs) = loop {
            let from_owner = self.router.owner_id_of(&from.id);
            let hits = match self.batcher.deliver(&publish.topic, from_owner, qos_retain).await {
                Ok(h) => h,
                Err(e) => {
                    // shared.rs:774-777: an Err of `matches` is logged and nobody is forwarded to
                    log::warn!("forwards, from:{:?}, topic:{:?}, error: {:?}", from, publish.topic, e);
                    return Ok(0);
                }
            };
            match self.deliver(&from, &publish, &hits) {
                Ok(done) => break done,
                Err(true) if tries < 3 => tries += 1,
                Err(_) => return self.inner.forwards(msg_id, from, publish).await, // $share members (or a table that will not hold still): the reference's own path
            }
        };
        let recipients_count = recipients.len();
        #[cfg(feature = "msgstore")]
        if let Some(msg_id) = msg_id {
            if !recipients.is_empty() {
                // shared.rs:790-797
                if let Err(e) = self.scx.extends.message_mgr().await.mark_forwarded(msg_id, recipients).await {
                    log::warn!("forwards: mark_forwarded error, msg_id: {:?}, {e}", msg_id);
                }
            }
        }
        if errs.is_empty() { Ok(recipients_count) } else { Err((recipients_count, errs)) }
    }

    // ---- everything else: plain delegation, as ClusterShared delegates to its `inner` -------------------------------------------
    async fn forwards_and_get_shareds(
        &self,
        from: From,
        publish: Publish,
    ) -> std::result::Result<(SubRelationsMap, ForwardedRecipients), (ForwardedRecipients, Vec<(To, From, Publish, Reason)>)> {
        self.inner.forwards_and_get_shareds(from, publish).await
    }
    async fn forwards_to(
        &self,
        from: From,
        publish: &Publish,
        relations: SubRelations,
        msg_id: Option<MsgID>,
    ) -> std::result::Result<ForwardedRecipients, (ForwardedRecipients, Vec<(To, From, Publish, Reason)>)> {
        self.inner.forwards_to(from, publish, relations, msg_id).await
    }
    fn iter(&self) -> Box<dyn Iterator<Item = Box<dyn Entry>> + Sync + Send + '_> { self.inner.iter() }
    fn random_session(&self) -> Option<Session> { self.inner.random_session() }
    async fn session_status(&self, client_id: &str) -> Option<SessionStatus> { self.inner.session_status(client_id).await }
    async fn client_states_count(&self) -> usize { self.inner.client_states_count().await }
    fn sessions_count(&self) -> usize { self.inner.sessions_count() }
    async fn query_subscriptions(&self, q: &SubsSearchParams) -> Vec<SubsSearchResult> { self.inner.query_subscriptions(q).await }
    async fn subscriptions_count(&self) -> usize { self.inner.subscriptions_count().await }
    #[cfg(feature = "msgstore")]
    async fn message_load(&self, client_id: &str, topic_filter: &str, group: Option<&SharedGroup>) -> Result<Vec<(MsgID, From, Publish)>> {
        self.inner.message_load(client_id, topic_filter, group).await
    }
    #[cfg(feature = "retain")]
    async fn retain_load_with(&self, topic_filter: &TopicFilter, cb: Arc<dyn rmqtt::shared::RetainLoadCallback>) -> Result<Vec<(NodeId, MsgID)>> {
        self.inner.retain_load_with(topic_filter, cb).await
    }
    #[cfg(feature = "msgstore")]
    async fn message_mark_forwarded(&self, from_node_id: NodeId, msg_id: MsgID, recipients: ForwardedRecipients) -> Result<()> {
        self.inner.message_mark_forwarded(from_node_id, msg_id, recipients).await
    }
}
