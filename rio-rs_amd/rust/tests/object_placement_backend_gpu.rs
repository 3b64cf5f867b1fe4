//! `gpu` twin of the provider conformance block of rio-rs/tests/object_placement_backend.rs:108-123.
//!
//! Goes into the reference tree as `rio-rs/tests/object_placement_backend_gpu.rs` (its own test binary, so the
//! reference's file stays untouched); the two generic checks are the reference's own (`no_placement`,
//! `save_and_load`, object_placement_backend.rs:11-34), restated here because a test binary cannot import another
//! one's private functions.  Run with
//!     RIO_GP_LIB_DIR=<dir of librio_gp.so> cargo test --features gpu --test object_placement_backend_gpu
//! on a machine with an MI355X.  NOT COMPILED IN THIS REPOSITORY (no cargo in the build image); the same assertions
//! run against the same C ABI through ctypes in tests/test_gpu_object_placement.py.
#![cfg(feature = "gpu")]

use rio_rs::{
    ObjectId,
    object_placement::{ObjectPlacement, ObjectPlacementItem, gpu::GpuObjectPlacement},
};

fn provider() -> GpuObjectPlacement {
    GpuObjectPlacement::builder()
        .max_objects(1 << 16)
        .max_nodes(64)
        .build()
        .expect("no gfx950 device: the provider has no CPU fallback")
}

// object_placement_backend.rs:11-16
async fn no_placement<S: ObjectPlacement>(provider: S) {
    provider.prepare().await.unwrap();
    let server_addr = provider.lookup(&ObjectId::new("obj", "1")).await.unwrap();
    assert!(server_addr.is_none());
}

// object_placement_backend.rs:18-34
async fn save_and_load<S: ObjectPlacement>(provider: S) {
    provider.prepare().await.unwrap();
    let obj_id = ObjectId::new("obj", "1");
    let placement = ObjectPlacementItem::new(obj_id, Some("0.0.0.0:8888".to_string()));
    provider.update(placement).await.unwrap();
    let server_addr = provider.lookup(&ObjectId::new("obj", "1")).await.unwrap();
    assert_eq!(server_addr.as_ref().unwrap(), "0.0.0.0:8888");
    provider.clean_server("0.0.0.0:8888".to_string()).await.unwrap();
    let server_addr = provider.lookup(&ObjectId::new("obj", "1")).await.unwrap();
    assert!(server_addr.is_none());
}

mod gpu {
    #[tokio::test]
    async fn no_placement() {
        super::no_placement(super::provider()).await;
    }

    #[tokio::test]
    async fn save_and_load() {
        super::save_and_load(super::provider()).await;
    }

    // the shared-clone usage of rio-rs/tests/server_utils.rs:62-73: many tasks, one provider, first requests after
    // a server joins (every task introduces its own address) — none may fail
    #[tokio::test(flavor = "multi_thread", worker_threads = 8)]
    async fn concurrent_first_requests_from_new_servers() {
        use rio_rs::object_placement::ObjectPlacementItem;
        use rio_rs::ObjectId;
        let provider = super::provider();
        let mut tasks = Vec::new();
        for t in 0..16 {
            let p = provider.clone();
            tasks.push(tokio::spawn(async move {
                for k in 0..50 {
                    let name = format!("{t}-{k}");
                    let addr = format!("10.0.{t}.{k}:5000");
                    // ObjectId is not Clone (service_object.rs:19-20): one for the update, one for the lookup
                    p.update(ObjectPlacementItem::new(ObjectId::new("obj", name.clone()), Some(addr.clone()))).await.unwrap();
                    assert_eq!(p.lookup(&ObjectId::new("obj", name)).await.unwrap().as_deref(), Some(addr.as_str()));
                }
            }));
        }
        for t in tasks {
            t.await.unwrap();
        }
    }
}
