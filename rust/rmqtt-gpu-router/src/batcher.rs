//! Deadline micro-batcher between `Router::matches` (one call per PUBLISH, from many tokio workers,
//! rmqtt/src/shared.rs:772) and the batched device pass (`rgr_group_match_batch_deliver`).
//!
//! Callers enqueue `(publisher owner id, topic, oneshot)` and await; ONE driver task drains the queue when it
//! holds `max_batch` publishes or `max_delay` has passed since the first one, runs one device pass on a
//! blocking thread and fans the per-topic hit slices back out.  No lock of the router is held while a caller
//! waits, and the FFI call never runs on a reactor thread.  Measured on MI355X (profiles/): 0.2 ms for a
//! batch of one, <1 ms for 4 096 publishes at config-2 fan-out — a 100-200 µs deadline costs little latency
//! and multiplies throughput.  C++ twin (compiled and tested): rmqtt_amd/host/gpu_router.cpp `Batcher`.
//! Source only (no rustc in the build image).
use std::sync::Arc;
use std::time::Duration;

use tokio::sync::{mpsc, oneshot};

use crate::ffi::*;

/// One publish's share of a device pass: `Err` = `Topic::from_str` failed (router.rs:177).
pub type Hits = Result<Vec<rgr_tuple>, String>;

pub struct MatchRequest {
    pub topic: String,
    /// dense owner id of the publisher's `Id` (RGR_ID_NONE when it holds no subscription): No Local is decided
    /// on the device against it (router.rs:196-201)
    pub from_owner: u32,
    pub reply: oneshot::Sender<Hits>,
}

/// `*mut rgr_group` that may cross threads: the C ABI is thread-safe (header, "threading").
#[derive(Clone, Copy)]
pub struct GroupPtr(pub *mut rgr_group);
unsafe impl Send for GroupPtr {}
unsafe impl Sync for GroupPtr {}

pub struct Batcher {
    tx: mpsc::UnboundedSender<MatchRequest>,
}

impl Batcher {
    /// `before_pass` runs on the blocking thread right before every device pass (the router commits pending
    /// subscription changes there, so `add`/`remove` never wait for the device).
    pub fn spawn<F>(g: GroupPtr, max_batch: usize, max_delay: Duration, before_pass: F) -> Self
    where
        F: Fn() -> Result<(), String> + Send + Sync + 'static,
    {
        let (tx, mut rx) = mpsc::unbounded_channel::<MatchRequest>();
        let before_pass = Arc::new(before_pass);
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
                let work: Vec<(String, u32)> = reqs.iter().map(|r| (r.topic.clone(), r.from_owner)).collect();
                let bp = before_pass.clone();
                let res = tokio::task::spawn_blocking(move || {
                    bp()?;
                    unsafe { match_many(g, &work) }
                })
                .await;
                match res {
                    Ok(Ok(per_topic)) => {
                        for (req, hits) in reqs.into_iter().zip(per_topic) {
                            let _ = req.reply.send(hits);
                        }
                    }
                    Ok(Err(e)) => reqs.into_iter().for_each(|r| { let _ = r.reply.send(Err(e.clone())); }),
                    Err(e) => reqs.into_iter().for_each(|r| { let _ = r.reply.send(Err(e.to_string())); }),
                }
            }
        });
        Self { tx }
    }

    pub async fn matches(&self, topic: &str, from_owner: u32) -> Hits {
        let (reply, rx) = oneshot::channel();
        self.tx.send(MatchRequest { topic: topic.to_owned(), from_owner, reply }).map_err(|e| e.to_string())?;
        rx.await.map_err(|e| e.to_string())?
    }
}

/// One device pass with the delivery stage (tuples carry RGR_HIT_* delivery words).  qos 2 / retain 0 as the
/// publish attributes leave the subscription's own qos in the word: `forwards_to` applies the publish's qos later
/// exactly as it does today (shared.rs:902).
unsafe fn match_many(g: GroupPtr, work: &[(String, u32)]) -> Result<Vec<Hits>, String> {
    let mut blob = Vec::new();
    let mut offs = vec![0u64];
    let mut attrs = Vec::with_capacity(work.len());
    for (t, owner) in work {
        blob.extend_from_slice(t.as_bytes());
        offs.push(blob.len() as u64);
        attrs.push(rgr_publish_attr { from_id: *owner, qos_retain: 2 });
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
                Ok(tuples[ho[i] as usize..ho[i + 1] as usize].to_vec())
            }
        })
        .collect();
    rgr_result_free(&mut res);
    Ok(out)
}
