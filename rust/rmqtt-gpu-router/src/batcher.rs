//! Deadline micro-batcher in front of `rgr_match_batch`.
//!
//! `Router::matches` is called once per PUBLISH from many tokio workers (rmqtt/src/shared.rs:772);
//! the GPU wants batches.  Callers enqueue (topic, oneshot) and await; one driver task drains the
//! queue when it holds `max_batch` topics or `max_delay` has elapsed since the first one, runs ONE
//! device pass and fans the per-topic tuple slices back out.  Measured on MI355X (host blob in,
//! host tuples out, config 2): 0.21 ms for a batch of 1, 0.68 ms for 4096, 2.0 ms for 200 000
//! (profiles/r01_latency_host_in_out_config2.txt) — so a 100-200 µs deadline adds little latency
//! while multiplying throughput.  Source only (no rustc in the build image).
use std::time::Duration;

use tokio::sync::{mpsc, oneshot};

use crate::ffi::*;

pub struct MatchRequest {
    pub topic: String,
    pub reply: oneshot::Sender<Result<Vec<rgr_tuple>, String>>,
}

pub struct Batcher {
    tx: mpsc::UnboundedSender<MatchRequest>,
}

impl Batcher {
    pub fn spawn(h: *mut rgr_handle, max_batch: usize, max_delay: Duration) -> Self {
        let (tx, mut rx) = mpsc::unbounded_channel::<MatchRequest>();
        let h = h as usize; // the handle is thread-safe; smuggle the pointer across the task boundary
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
                // one device pass for the whole batch (blocking FFI: run it off the reactor)
                let topics: Vec<String> = reqs.iter().map(|r| r.topic.clone()).collect();
                let res = tokio::task::spawn_blocking(move || unsafe { match_many(h as *mut rgr_handle, &topics) }).await;
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

    pub async fn matches(&self, topic: &str) -> Result<Vec<rgr_tuple>, String> {
        let (reply, rx) = oneshot::channel();
        self.tx.send(MatchRequest { topic: topic.to_owned(), reply }).map_err(|e| e.to_string())?;
        rx.await.map_err(|e| e.to_string())?
    }
}

/// Err(invalid topic) per topic mirrors `Topic::from_str` failing in `_matches` (router.rs:177).
unsafe fn match_many(h: *mut rgr_handle, topics: &[String]) -> Result<Vec<Result<Vec<rgr_tuple>, String>>, String> {
    let mut blob = Vec::new();
    let mut offs = vec![0u64];
    for t in topics {
        blob.extend_from_slice(t.as_bytes());
        offs.push(blob.len() as u64);
    }
    let mut res: rgr_result = std::mem::zeroed();
    if rgr_match_batch(h, blob.as_ptr(), offs.as_ptr(), topics.len() as u32, &mut res) != RGR_OK {
        return Err(std::ffi::CStr::from_ptr(rgr_last_error()).to_string_lossy().into_owned());
    }
    let status = std::slice::from_raw_parts(res.status, topics.len());
    let ho = std::slice::from_raw_parts(res.hit_offsets, topics.len() + 1);
    let tuples = std::slice::from_raw_parts(res.tuples, res.n_hits as usize);
    let out = (0..topics.len())
        .map(|i| {
            if status[i] != RGR_TOPIC_OK {
                Err(format!("invalid topic `{}`", topics[i]))
            } else {
                Ok(tuples[ho[i] as usize..ho[i + 1] as usize].to_vec())
            }
        })
        .collect();
    rgr_result_free(&mut res);
    Ok(out)
}
